// afec_amd/csrc/afx_workspace.cpp -- the device side a batch borrows (pooled workspaces: buffers, streams, events, the
// work queues' counters), how the host waits for the device, the plan's two copy streams, and page-locked host memory.
// See afx_host.h for the map of the host side.

#include <sys/mman.h>

#include <algorithm>
#include <chrono>
#include <new>
#include <thread>

#include "afx_host.h"

namespace afx {
namespace host {

std::vector<Workspace::Buf*> Workspace::all_bufs() {
  return {&tables, &pcm, &rec, &mag, &stats, &follower, &efflen, &raw, &files, &scan, &partial, &place, &queue, &rt_polar, &rt_odf,
          &rt_onsets, &rt_scratch, &rt_scalars, &rt_stats, &stat_tmp, &rs_files, &rs_groups, &rs_ngroups};
}
size_t Workspace::bytes() {
  size_t n = 0;
  for (Buf* b : all_bufs()) n += b->cap;
  return n;
}

void ws_free(Workspace* w) {
  if (!w) return;
  for (Workspace::Buf* b : w->all_bufs()) hipFree(b->p);
  if (w->h_pin) hipHostFree(w->h_pin);
  if (w->ev0) hipEventDestroy(w->ev0);
  if (w->ev1) hipEventDestroy(w->ev1);
  if (w->ev_fork) hipEventDestroy(w->ev_fork);
  if (w->ev_join) hipEventDestroy(w->ev_join);
  if (w->ev_time_join) hipEventDestroy(w->ev_time_join);
  if (w->ev_copy) hipEventDestroy(w->ev_copy);
  if (w->side_stream) hipStreamDestroy(w->side_stream);
  if (w->stream) hipStreamDestroy(w->stream);
  delete w;
}

// pooled workspaces: at most 16 idle ones per plan, none larger than 4 GiB (a 1024-file batch of one-second files with
// every descriptor needs ~1 GiB: magnitudes 8 KiB and PCM 8 KiB per frame)
constexpr size_t kPoolMaxIdle = 16;
constexpr size_t kPoolMaxBytes = (size_t)4 << 30;

Workspace* ws_acquire(afx_plan* plan, hipError_t* err) {
  {
    std::lock_guard<std::mutex> lock(plan->pool_mutex);
    if (!plan->pool.empty()) {
      Workspace* w = plan->pool.back();
      plan->pool.pop_back();
      w->blocking = plan->blocking_wait.load();
      return w;
    }
  }
  Workspace* w = new (std::nothrow) Workspace();
  if (!w) { *err = hipErrorOutOfMemory; return nullptr; }
  hipError_t e = hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreate(&w->ev0);
  if (e == hipSuccess) e = hipEventCreate(&w->ev1);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&w->side_stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&w->ev_fork, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&w->ev_join, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&w->ev_time_join, hipEventDisableTiming);
  w->blocking = plan->blocking_wait.load();
  if (e == hipSuccess) e = hipEventCreateWithFlags(&w->ev_copy, hipEventDisableTiming);
  if (e != hipSuccess) { *err = e; ws_free(w); return nullptr; }
  return w;
}

void ws_release(afx_plan* plan, Workspace* w) {
  if (!w) return;
  if (w->bytes() <= kPoolMaxBytes) {
    std::lock_guard<std::mutex> lock(plan->pool_mutex);
    if (plan->pool.size() < kPoolMaxIdle) {
      plan->pool.push_back(w);
      return;
    }
  }
  ws_free(w);
}

hipError_t ws_pin_reserve(Workspace* w, size_t bytes) {
  if (bytes <= w->h_pin_cap) return hipSuccess;
  if (w->h_pin) hipHostFree(w->h_pin);
  w->h_pin = nullptr;
  w->h_pin_cap = 0;
  const size_t want = bytes + bytes / 2 + 4096;
  const hipError_t e = hipHostMalloc(&w->h_pin, want, hipHostMallocDefault);
  if (e != hipSuccess) { w->h_pin = nullptr; return e; }
  w->h_pin_cap = want;
  return hipSuccess;
}

size_t pool_trim(afx_plan* plan) {
  std::vector<Workspace*> idle;
  {
    std::lock_guard<std::mutex> lock(plan->pool_mutex);
    idle.swap(plan->pool);
  }
  size_t freed = 0;
  for (Workspace* w : idle) {
    freed += w->bytes();
    ws_free(w);
  }
  return freed;
}

hipError_t ws_reserve(afx_plan* plan, Workspace::Buf& b, size_t bytes) {
  if (bytes <= b.cap) return hipSuccess;
  hipFree(b.p);
  b.p = nullptr;
  b.cap = 0;
  size_t want = bytes + bytes / 4 + 4096;
  hipError_t e = hipMalloc(&b.p, want);
  if (e == hipErrorOutOfMemory) {
    // The idle workspaces of the pool keep their capacity (up to 16 of them, up to 4 GiB each): what this batch lacks may
    // be lying there.  Give it back and ask once more, without the growth margin.
    (void)hipGetLastError();
    b.p = nullptr;
    if (pool_trim(plan) > 0) {
      want = bytes;
      e = hipMalloc(&b.p, want);
    }
  }
  if (e != hipSuccess) { b.p = nullptr; return e; }
  b.cap = want;
  return hipSuccess;
}

// The host's waits.  hipStreamSynchronize / hipEventSynchronize spin (one busy CPU per waiting thread); with
// afx_plan_set_blocking_wait the thread polls the workspace's copy event and sleeps in between.  The runtime's own
// alternatives did not serve: hipEventBlockingSync alone changes nothing here (the eight workers of a crawl still keep
// 6.6 CPUs busy), and hipDeviceScheduleBlockingSync, set on a device that is already active, left a later
// hipStreamSynchronize hanging.
// (600: round 5, tools/thread_cpu.py on the C4 share -- 5 workers 1.84 -> 1.78 busy CPUs at 284 -> 287 k files/s, 8 workers
// 1.99 -> 1.90; 1 000 us starts to cost throughput with few workers)
constexpr int kNapCeilingUs = 600;
hipError_t wait_for_event(Workspace* ws, hipEvent_t ev) {
  if (!ws->blocking) return hipEventSynchronize(ev);
  // naps grow from 20 us to kNapCeilingUs: what a crawl waits for takes milliseconds (an upload 1.6 ms, a batch's
  // kernels 10+ ms), other batches are in flight meanwhile, and every poll is a system call + a wake-up
  for (int nap = 20;; nap = std::min(nap + nap / 2, kNapCeilingUs)) {
    const hipError_t e = hipEventQuery(ev);
    if (e != hipErrorNotReady) return e;
    std::this_thread::sleep_for(std::chrono::microseconds(nap));
  }
}
hipError_t wait_for_stream(Workspace* ws, hipStream_t stream) {
  if (!ws || !ws->blocking) return hipStreamSynchronize(stream);
  hipError_t e = hipEventRecord(ws->ev_copy, stream);
  return e == hipSuccess ? wait_for_event(ws, ws->ev_copy) : e;
}

// One large host-to-device transfer through the plan's upload stream; returns when it has landed.  The wait is the
// host's, not a hipStreamWaitEvent of the batch's stream: HIP multiplexes its streams onto a few hardware queues
// (GPU_MAX_HW_QUEUES, 4 by default), and a barrier packet that waits for a 1.6 ms upload stalls every other stream
// that shares the queue -- with six batches in flight the kernels of one batch and the uploads of the next then
// exclude each other (measured: kernels busy 0.55, copies busy 0.59, either busy 0.85 of a crawl).
hipError_t upload_through_plan(afx_plan* plan, Workspace* ws, void* dst, const void* src, size_t bytes) {
  {
    std::lock_guard<std::mutex> lock(plan->up_mutex);
    hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, plan->up_stream);
    if (e == hipSuccess) e = hipEventRecord(ws->ev_copy, plan->up_stream);
    if (e != hipSuccess) return e;
  }
  return wait_for_event(ws, ws->ev_copy);
}
// Device-to-host transfers of a batch's results through the plan's download stream, behind everything enqueued on the
// batch's stream so far (waited for on the host, for the same reason as above); returns when they have landed.
hipError_t download_through_plan(afx_batch* b, const Download* items, int n) {
  afx_plan* plan = b->plan;
  Workspace* ws = b->ws;
  hipError_t e = wait_for_stream(ws, b->stream);
  if (e != hipSuccess) return e;
  {
    std::lock_guard<std::mutex> lock(plan->down_mutex);
    for (int i = 0; i < n && e == hipSuccess; ++i)
      if (items[i].dst && items[i].bytes) e = hipMemcpyAsync(items[i].dst, items[i].src, items[i].bytes, hipMemcpyDeviceToHost, plan->down_stream);
    if (e == hipSuccess) e = hipEventRecord(ws->ev_copy, plan->down_stream);
    if (e != hipSuccess) return e;
  }
  return wait_for_event(ws, ws->ev_copy);
}

}  // namespace host
}  // namespace afx

using namespace afx::host;

// Page-locked host memory.  Large blocks are anonymous mappings on 2 MiB boundaries with MADV_HUGEPAGE, touched and then
// registered with the runtime: page-locking huge pages takes a third of hipHostMalloc's time (7.3 vs 21-23 ms per
// 128 MiB on the MI355X box, tools/pin_rate.py) and the first crawl of a process locks ~1.2 GB.  Small blocks, and
// large ones when the mapping or the registration fails, come from hipHostMalloc.
namespace {
std::mutex g_host_mutex;
std::vector<std::pair<void*, size_t>> g_host_mapped;   // registered mappings: base, mapped bytes
constexpr size_t kHugePage = (size_t)2 << 20;
}  // namespace

extern "C" {

void* afx_host_alloc(int64_t bytes) {
  if (bytes <= 0) return nullptr;
  if ((size_t)bytes >= 4 * kHugePage) {
    const size_t len = ((size_t)bytes + kHugePage - 1) & ~(kHugePage - 1), mapped = len + kHugePage;
    void* raw = mmap(nullptr, mapped, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (raw != MAP_FAILED) {
      // trim to a 2 MiB boundary so that the whole block can be backed by huge pages
      char* base = (char*)(((uintptr_t)raw + kHugePage - 1) & ~(uintptr_t)(kHugePage - 1));
      if (base > (char*)raw) munmap(raw, (size_t)(base - (char*)raw));
      const size_t tail = (size_t)((char*)raw + mapped - (base + len));
      if (tail) munmap(base + len, tail);
      madvise(base, len, MADV_HUGEPAGE);
      for (size_t off = 0; off < len; off += 4096) base[off] = 0;   // fault the pages in before they are locked
      // portable: the streaming driver's pools hand a buffer to workers of any device
      if (hipHostRegister(base, len, hipHostRegisterPortable) == hipSuccess) {
        std::lock_guard<std::mutex> lock(g_host_mutex);
        g_host_mapped.emplace_back(base, len);
        return base;
      }
      (void)hipGetLastError();
      munmap(base, len);
    }
  }
  void* p = nullptr;
  if (hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
  return p;
}
void afx_host_free(void* p) {
  if (!p) return;
  size_t len = 0;
  {
    std::lock_guard<std::mutex> lock(g_host_mutex);
    for (size_t i = 0; i < g_host_mapped.size(); ++i)
      if (g_host_mapped[i].first == p) {
        len = g_host_mapped[i].second;
        g_host_mapped.erase(g_host_mapped.begin() + (long)i);
        break;
      }
  }
  if (len) {
    hipHostUnregister(p);
    munmap(p, len);
  } else hipHostFree(p);
}

}  // extern "C"
