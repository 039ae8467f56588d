// tests/sanitize/hipstub/hip_stub.cpp -- see hip/hip_runtime.h beside it: a host-memory stand-in for the HIP runtime,
// TEST INFRASTRUCTURE ONLY.  Everything is synchronous (a copy has happened when the call returns, a "kernel" -- the
// mock launchers -- has run), streams and events are heap objects so that a leaked one is a leak the sanitizer sees,
// and the device can be made to run out of memory or to go away.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

struct ihipStream_t { int device; };
struct ihipEvent_t { int device; std::atomic<long long> ns{0}; };

namespace {
std::mutex g_mutex;
int g_devices = 1;
thread_local int t_device = 0;
size_t g_limit = 0;
std::atomic<long> g_fail_after{-1};
std::map<void*, std::pair<size_t, int>> g_blocks;     // device allocations: bytes, device
size_t g_in_use[64] = {};
bool g_lost[64] = {};
std::atomic<long> g_streams{0}, g_events{0};
thread_local hipError_t t_last = hipSuccess;

hipError_t done(hipError_t e) { if (e != hipSuccess) t_last = e; return e; }
bool lost() { return t_device >= 0 && t_device < 64 && g_lost[t_device]; }
#define STUB_LIVE() do { if (lost()) return done(hipErrorUnknown); } while (0)
long long now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

namespace hipstub {
void set_device_count(int n) { std::lock_guard<std::mutex> l(g_mutex); g_devices = n; }
void set_memory_limit(size_t bytes) { std::lock_guard<std::mutex> l(g_mutex); g_limit = bytes; }
void fail_allocation_after(long n) { g_fail_after = n; }
void lose_device(int device, bool is_lost) { std::lock_guard<std::mutex> l(g_mutex); if (device >= 0 && device < 64) g_lost[device] = is_lost; }
size_t device_bytes_in_use() { std::lock_guard<std::mutex> l(g_mutex); size_t n = 0; for (size_t v : g_in_use) n += v; return n; }
long live_streams() { return g_streams.load(); }
long live_events() { return g_events.load(); }
}  // namespace hipstub

extern "C" {

hipError_t hipGetDeviceCount(int* count) { std::lock_guard<std::mutex> l(g_mutex); *count = g_devices; return g_devices > 0 ? hipSuccess : done(hipErrorNoDevice); }
hipError_t hipSetDevice(int device) {
  std::lock_guard<std::mutex> l(g_mutex);
  if (device < 0 || device >= g_devices) return done(hipErrorInvalidDevice);
  t_device = device;
  return hipSuccess;
}
hipError_t hipGetDevice(int* device) { *device = t_device; return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* prop, int device) {
  std::lock_guard<std::mutex> l(g_mutex);
  if (device < 0 || device >= g_devices) return done(hipErrorInvalidDevice);
  std::memset(prop, 0, sizeof *prop);
  std::strcpy(prop->name, "mock device (tests/sanitize/hipstub)");
  std::strcpy(prop->gcnArchName, "gfx950:mock");
  prop->totalGlobalMem = (size_t)288 << 30;
  prop->multiProcessorCount = 256;
  return hipSuccess;
}
hipError_t hipDeviceGetAttribute(int* value, hipDeviceAttribute_t, int) { *value = 256; return hipSuccess; }
hipError_t hipGetLastError(void) { const hipError_t e = t_last; t_last = hipSuccess; return e; }
const char* hipGetErrorString(hipError_t e) {
  switch (e) {
    case hipSuccess: return "no error";
    case hipErrorOutOfMemory: return "out of memory";
    case hipErrorNotReady: return "not ready";
    case hipErrorInvalidDevice: return "invalid device ordinal";
    case hipErrorNoDevice: return "no device";
    default: return "mock device error";
  }
}

hipError_t hipMalloc(void** p, size_t bytes) {
  *p = nullptr;
  STUB_LIVE();
  if (g_fail_after.load() >= 0 && g_fail_after.fetch_sub(1) == 0) return done(hipErrorOutOfMemory);
  std::lock_guard<std::mutex> l(g_mutex);
  if (g_limit && g_in_use[t_device] + bytes > g_limit) return done(hipErrorOutOfMemory);
  void* q = std::malloc(bytes ? bytes : 1);
  if (!q) return done(hipErrorOutOfMemory);
  // fresh device memory holds anything: poison it, so that a kernel argument the host forgot to initialise shows
  std::memset(q, 0xA5, bytes);
  g_blocks[q] = {bytes, t_device};
  g_in_use[t_device] += bytes;
  *p = q;
  return hipSuccess;
}
hipError_t hipFree(void* p) {
  if (!p) return hipSuccess;
  std::lock_guard<std::mutex> l(g_mutex);
  auto it = g_blocks.find(p);
  if (it == g_blocks.end()) return done(hipErrorInvalidValue);
  g_in_use[it->second.second] -= it->second.first;
  g_blocks.erase(it);
  std::free(p);
  return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned) { *p = std::malloc(bytes ? bytes : 1); return *p ? hipSuccess : done(hipErrorOutOfMemory); }
hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
hipError_t hipHostRegister(void*, size_t, unsigned) { return hipSuccess; }
hipError_t hipHostUnregister(void*) { return hipSuccess; }
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind) { STUB_LIVE(); if (bytes) std::memmove(dst, src, bytes); return hipSuccess; }
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind, hipStream_t) { STUB_LIVE(); if (bytes) std::memmove(dst, src, bytes); return hipSuccess; }
hipError_t hipMemset(void* dst, int value, size_t bytes) { STUB_LIVE(); if (bytes) std::memset(dst, value, bytes); return hipSuccess; }
hipError_t hipMemsetAsync(void* dst, int value, size_t bytes, hipStream_t) { STUB_LIVE(); if (bytes) std::memset(dst, value, bytes); return hipSuccess; }

hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { STUB_LIVE(); *s = new ihipStream_t{t_device}; ++g_streams; return hipSuccess; }
hipError_t hipStreamCreate(hipStream_t* s) { return hipStreamCreateWithFlags(s, 0); }
hipError_t hipStreamDestroy(hipStream_t s) { if (s) { delete s; --g_streams; } return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { STUB_LIVE(); return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { STUB_LIVE(); return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { STUB_LIVE(); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { STUB_LIVE(); *e = new ihipEvent_t{t_device}; ++g_events; return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { return hipEventCreateWithFlags(e, 0); }
hipError_t hipEventDestroy(hipEvent_t e) { if (e) { delete e; --g_events; } return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { STUB_LIVE(); e->ns = now_ns(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { STUB_LIVE(); return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { STUB_LIVE(); return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)((double)(b->ns.load() - a->ns.load()) * 1e-6) + 1e-3f; return hipSuccess; }

}  // extern "C"

// for a ctypes caller (tests/test_sanitize_cpu.py loads the mock host library into Python)
extern "C" void hipstub_set_device_count(int n) { hipstub::set_device_count(n); }
