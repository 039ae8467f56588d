// tools/clock_probe/clock_probe.hip -- measurement aid of bench.py (not part of the product library and not on any
// product path): the shader clock the GPU sustains WHILE the timed launches run.
//
// One 64-lane wave on a stream of its own reads the shader-clock counter (s_memtime) and the constant-rate counter
// (s_memrealtime, hipDeviceAttributeWallClockRate kHz) every ~25 us, sleeping (s_sleep) in between, until the host
// sets a flag in mapped host memory -- or until its own time limit, so that a host that went away never leaves a
// kernel spinning.  clock = d(s_memtime) / d(s_memrealtime) x wall-clock rate.  The f64 kernels of this repository
// run at 1.9-2.1 of the part's 2.4 GHz depending on the box and the load (frames32 kernel): a VALU
// ceiling computed from another run's clock is not evidence for this one.
//
// The probe wave is resident before the first timed launch (start() waits for its first sample) and occupies one wave
// slot of one SIMD; it issues a handful of scalar instructions per 25 us.
//
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC clock_probe.hip -o ../../afec_amd/lib/libafx_clock_probe.so
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

namespace {

struct Shared {            // mapped, coherent host memory: the kernel's view of the host and back
  volatile int stop;       // host -> kernel
  volatile int started;    // kernel -> host: the first sample has been taken
  volatile int ended_by;   // kernel -> host: 1 = the stop flag, 2 = the time limit
  volatile int count;      // samples written
};

__global__ void clock_probe_kernel(Shared* sh, uint64_t* samples, int capacity, uint64_t max_real_ticks) {
  uint64_t core0, real0;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(core0), "=s"(real0));
  int n = 0, ended = 0;
  for (;;) {
    uint64_t core, real;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(core), "=s"(real));
    if (threadIdx.x == 0 && n < capacity) {
      samples[2 * n] = core;
      samples[2 * n + 1] = real;
    }
    if (n < capacity) ++n;
    if (n == 1 && threadIdx.x == 0) __hip_atomic_store(&sh->started, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (real - real0 > max_real_ticks) { ended = 2; break; }
    if (__hip_atomic_load(&sh->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) { ended = 1; break; }
    // ~25 us: s_sleep 127 = 127 x 64 clocks
    __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127);
    __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127);
  }
  if (threadIdx.x == 0) {
    // the closing sample (the loop's last one may be ~25 us old only when the limit ended it; take a fresh one anyway)
    uint64_t core, real;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(core), "=s"(real));
    const int last = n < capacity ? n : capacity - 1;
    samples[2 * last] = core;
    samples[2 * last + 1] = real;
    __hip_atomic_store(&sh->count, last + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&sh->ended_by, ended, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

struct Probe {
  int device = 0;
  hipStream_t stream = nullptr;
  Shared* shared = nullptr;        // host pointer (hipHostMalloc, mapped)
  uint64_t* d_samples = nullptr;
  int capacity = 0;
  double wall_khz = 0;
};

void destroy(Probe* p) {
  if (!p) return;
  if (p->d_samples) (void)hipFree(p->d_samples);
  if (p->shared) (void)hipHostFree(p->shared);
  if (p->stream) (void)hipStreamDestroy(p->stream);
  delete p;
}

}  // namespace

extern "C" {

// Starts the probe wave on `device`; it ends by afx_clock_probe_stop or after max_seconds on its own.  Returns NULL when
// the probe could not be started (no message is an error of the measurement, never of the product).
void* afx_clock_probe_start(int device, double max_seconds) {
  if (hipSetDevice(device) != hipSuccess) return nullptr;
  Probe* p = new Probe;
  p->device = device;
  int khz = 0;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) != hipSuccess || khz <= 0) khz = 100000;
  p->wall_khz = (double)khz;
  if (max_seconds < 0.01) max_seconds = 0.01;
  if (max_seconds > 20.0) max_seconds = 20.0;
  p->capacity = (int)(max_seconds / 20e-6) + 64;
  if (hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking) != hipSuccess ||
      hipHostMalloc((void**)&p->shared, sizeof(Shared), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
      hipMalloc((void**)&p->d_samples, (size_t)p->capacity * 2 * sizeof(uint64_t)) != hipSuccess) {
    (void)hipGetLastError();
    destroy(p);
    return nullptr;
  }
  p->shared->stop = 0; p->shared->started = 0; p->shared->ended_by = 0; p->shared->count = 0;
  Shared* d_shared = nullptr;
  if (hipHostGetDevicePointer((void**)&d_shared, p->shared, 0) != hipSuccess) { destroy(p); return nullptr; }
  const uint64_t max_ticks = (uint64_t)(max_seconds * p->wall_khz * 1e3);
  hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, p->stream, d_shared, p->d_samples, p->capacity, max_ticks);
  if (hipGetLastError() != hipSuccess) { destroy(p); return nullptr; }
  // resident before the caller's first launch: wait for the first sample (at most a second)
  const auto t0 = std::chrono::steady_clock::now();
  while (!p->shared->started) {
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 1.0) {
      p->shared->stop = 1;
      (void)hipStreamSynchronize(p->stream);
      destroy(p);
      return nullptr;
    }
    std::this_thread::sleep_for(std::chrono::microseconds(20));
  }
  return p;
}

// Ends the probe and reports: out[0] mean clock in GHz over the probe's life, out[1] / out[2] the lowest / highest clock
// over windows of >= 1 ms, out[3] samples, out[4] seconds covered, out[5] how it ended (1 = by this call, 2 = by its own
// time limit: then the caller's launches may have queued behind it and the timing is not to be used), out[6] the
// wall-clock counter's rate in kHz.  Returns 0, or -1 when the probe's samples could not be read.
int afx_clock_probe_stop(void* probe, double* out /* [7] */) {
  Probe* p = (Probe*)probe;
  if (!p) return -1;
  (void)hipSetDevice(p->device);
  p->shared->stop = 1;
  int rc = -1;
  if (hipStreamSynchronize(p->stream) == hipSuccess) {
    const int n = p->shared->count;
    std::vector<uint64_t> s((size_t)(n > 0 ? n : 0) * 2);
    if (n >= 2 && hipMemcpy(s.data(), p->d_samples, s.size() * sizeof(uint64_t), hipMemcpyDeviceToHost) == hipSuccess) {
      const double hz = p->wall_khz * 1e3;
      const double span_real = (double)(s[2 * (size_t)(n - 1) + 1] - s[1]);
      out[0] = span_real > 0 ? (double)(s[2 * (size_t)(n - 1)] - s[0]) / span_real * hz * 1e-9 : 0.0;
      double lo = 1e30, hi = 0.0;
      size_t a = 0;
      const double window = 1e-3 * hz;
      for (size_t b = 1; b < (size_t)n; ++b) {
        if ((double)(s[2 * b + 1] - s[2 * a + 1]) >= window) {
          const double g = (double)(s[2 * b] - s[2 * a]) / (double)(s[2 * b + 1] - s[2 * a + 1]) * hz * 1e-9;
          if (g < lo) lo = g;
          if (g > hi) hi = g;
          a = b;
        }
      }
      if (hi == 0.0) lo = hi = out[0];
      out[1] = lo; out[2] = hi; out[3] = (double)n; out[4] = span_real / hz; out[5] = (double)p->shared->ended_by; out[6] = p->wall_khz;
      rc = 0;
    }
  }
  (void)hipGetLastError();
  destroy(p);
  return rc;
}

}  // extern "C"
