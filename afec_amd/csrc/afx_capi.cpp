// afec_amd/csrc/afx_capi.cpp -- host side of libafx_hip.so: plans, batches and the C-ABI of
// include/afx.h.  Everything that computes a descriptor runs in afx_kernels.hip on the GPU; this
// file only builds the constant tables (the analogue of the TSampleAnalyser constructor,
// SampleAnalyser.cpp:162-198), packs buffers into an HBM arena, cuts the frame loop
// (SampleAnalyser.cpp:760-764, 814) into per-wave chunks and moves results.

#include "../../include/afx.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>

#include <sys/mman.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "afx_internal.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

namespace {

thread_local std::string g_last_error;
#if defined(AFX_STAMPS) && AFX_STAMPS
unsigned long long* g_stamp_buf = nullptr;
#endif

int fail(int status, const std::string& msg) {
  g_last_error = msg;
  return status;
}
int hip_fail(hipError_t e, const char* what) {
  (void)hipGetLastError();   // the runtime keeps the last error per thread: a later hipGetLastError() check must not see this one
  return fail(e == hipErrorOutOfMemory ? AFX_ERR_OUT_OF_MEMORY : AFX_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(expr)                                  \
  do {                                                 \
    hipError_t e_ = (expr);                            \
    if (e_ != hipSuccess) return hip_fail(e_, #expr);  \
  } while (0)

// TMath::d2iRound (CoreTypes/Export/InlineMath.inl:823-826): truncate(x + sign(x)/2)
int d2i_round(double v) { return (int)(v + ((v < 0.0) ? -0.5 : 0.5)); }
// TAudioMath::MsToSamples (AudioTypes/Export/AudioMath.inl:125-128), float arithmetic
int ms_to_samples(int rate, float ms) {
  const float v = (float)rate / 1000.0f * ms;
  return (int)(v + ((v < 0.0f) ? -0.5f : 0.5f));
}

// Hann window of LibXtract (window.c:67-76: denominator N-1), times 2 (SampleAnalyser.cpp:178-181)
std::vector<double> build_window(int n) {
  std::vector<double> w(n);
  const double M = n - 1;
  for (int i = 0; i < n; ++i) w[i] = 0.5 * (1.0 - std::cos(2.0 * M_PI * (double)i / M));
  for (int i = 0; i < n; ++i) w[i] *= 2.0;
  return w;
}

// Mel filter bank exactly as xtract_init_mfcc builds it for XTRACT_EQUAL_GAIN (LibXtract
// init.c:237-382), called with N = fft/2 and nyquist = sample_rate/2 (SampleAnalyser.cpp:195-197):
// peaks are placed with M = N >> 1 and truncated to int, the first rise divides by fft_peak[0] == 0
// and the bin cursor runs on from one filter into the next.
std::vector<double> build_mel(int N, double nyquist, double fmin, double fmax, int nb) {
  std::vector<double> tab((size_t)nb * N, 0.0);
  const double mel_hi = 1127 * std::log(1 + fmax / 700);
  const double mel_lo = 1127 * std::log(1 + fmin / 700);
  const double step = (mel_hi - mel_lo) / nb;
  std::vector<double> mel(nb + 2), lin(nb + 2);
  std::vector<int> peak(nb + 2);
  const int M = N >> 1;
  mel[0] = mel_lo;
  lin[0] = fmin;
  peak[0] = (int)(lin[0] / nyquist * M);
  for (int n = 1; n < nb + 2; ++n) {
    mel[n] = mel[n - 1] + step;
    lin[n] = 700 * (std::exp(mel[n] / 1127) - 1);
    peak[n] = (int)(lin[n] / nyquist * M);
  }
  int cursor = 0;
  for (int n = 0; n < nb; ++n) {
    double* row = tab.data() + (size_t)n * N;
    const double height = 1.0;
    double inc = (n == 0) ? height / peak[n] : height / (peak[n] - peak[n - 1]);
    double val = 0;
    for (int k = 0; k < cursor; ++k) row[k] = 0.0;
    for (; cursor <= peak[n]; ++cursor) {
      row[cursor] = val;
      val += inc;
    }
    inc = height / (peak[n + 1] - peak[n]);
    val = 0;
    for (cursor = peak[n + 1]; cursor > peak[n]; --cursor) {
      row[cursor] = val;
      val += inc;
    }
    for (int k = peak[n + 1] + 1; k < N; ++k) row[k] = 0.0;
  }
  return tab;
}

template <typename T>
struct cpx {
  T re, im;
};

// e^{-2 pi i e / n}, evaluated in long double and reduced to the first octant for accuracy
template <typename T>
cpx<T> twiddle(long long e, long long n) {
  e %= n;
  if (e < 0) e += n;
  const long double ang = -2.0L * 3.14159265358979323846264338327950288L * (long double)e / (long double)n;
  return {(T)std::cos(ang), (T)std::sin(ang)};
}

struct DeviceTables {
  void* win = nullptr;
  void* t1 = nullptr;
  void* t2 = nullptr;
  void* post = nullptr;
  void* melw = nullptr;
  double* dct = nullptr;
  // double-precision FFT tables for the time-domain neighbours (alias t1/t2/post in the f64 mode)
  void* t1_f64 = nullptr;
  void* t2_f64 = nullptr;
  void* post_f64 = nullptr;
  // half-wave kernels (afx_frames32.hip), always double
  void* win32 = nullptr;
  void* tw32 = nullptr;
  void* post32 = nullptr;
  void* melw32 = nullptr;
  // rhythm tracker (afx_rhythm.hip)
  double* rt_window = nullptr;
  double* rt_tw = nullptr;
  double* rt_ut = nullptr;
  double* rt_canny = nullptr;
  double* rt_rayleigh = nullptr;
  // sample-rate conversion (afx_resample.hip): uploaded by the first batch that holds a file at another rate
  float* rs_filter = nullptr;
};

}  // namespace

// Device buffers, stream and events of one batch.  Kept in a small per-plan pool so that the
// one-file-per-call pattern of the reference (one Extract() per worker thread and file,
// Crawler.cpp:706-728) does not pay hipMalloc / hipStreamCreate on every call.
struct Workspace {
  struct Buf {
    void* p = nullptr;
    size_t cap = 0;
  };
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // second stream for the rhythm tracker's kernels, which depend on the PCM only: small batches (a crawler's 256 files)
  // leave most of the chip idle during any one kernel, so the two kernel chains run side by side
  hipStream_t side_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipEvent_t ev_time_join = nullptr;   // the time-domain kernels (autocorrelation, f0, hop) run on the side stream too
  hipEvent_t ev_copy = nullptr;   // the host's waits for the plan's upload / download streams (and, when blocking, for this batch's stream)
  bool blocking = false;          // the host's waits of this workspace's batches sleep (afx_plan_set_blocking_wait)
  unsigned queue_count[afx::kQueueSlots] = {};   // values of the device work-queue counters after the launches enqueued so far
  bool queue_dirty = false;                      // a run failed between counting a launch and enqueuing it: reset both before the next
  Buf pcm, chunks, wchunks, rem, rec, mag, foff, stats, cfirst, follower, spans, efflen, raw, files, scan, partial, place, queue;
  Buf rt_files, rt_odf, rt_onsets, rt_scratch, rt_scalars, rt_stats, rt_foff, rt_long, rt_polar;   // rhythm tracker
  Buf stat_tmp;                                                                 // half-wave statistics class
  Buf rs_files, rs_groups, rs_ngroups;                                          // sample-rate conversion (afx_resample.hip)
  size_t bytes() const {
    return pcm.cap + chunks.cap + wchunks.cap + rem.cap + rec.cap + mag.cap + foff.cap + stats.cap + cfirst.cap + follower.cap + spans.cap +
           efflen.cap + raw.cap + files.cap + scan.cap + partial.cap + place.cap + queue.cap + rt_files.cap + rt_long.cap + rt_polar.cap + rt_odf.cap +
           rt_onsets.cap + rt_scratch.cap + rt_scalars.cap + rt_stats.cap + rt_foff.cap + stat_tmp.cap + rs_files.cap + rs_groups.cap + rs_ngroups.cap;
  }
};

struct afx_plan {
  // the plan handle and every live batch hold one reference; the last one to go frees the plan (a batch destroyed
  // after afx_plan_destroy still finds its plan, its device and its workspace pool)
  std::atomic<int> refs{1};
  std::mutex pool_mutex;
  std::atomic<bool> blocking_wait{false};   // afx_plan_set_blocking_wait: the host's waits of this plan's batches sleep
  std::vector<Workspace*> pool;
  afx_plan_desc desc;
  int first_bin, last_bin, bin_count;
  std::vector<double> window;  // [fft]
  std::vector<double> mel;     // [14][fft/2]
  DeviceTables dev;
  int cu_count = 256;
  // The large transfers of every batch of this plan go through ONE upload stream and ONE download stream.  Measured
  // on the pool (tools/link_rate.py): one stream per direction runs full duplex at 46 + 46 GB/s, three batch streams
  // that each upload and download collapse to 16 + 16 GB/s (the copy engines are re-assigned back and forth).
  std::mutex up_mutex, down_mutex;
  hipStream_t up_stream = nullptr, down_stream = nullptr;
  // afx_plan_desc.frame_kernel: 0 = 64-lane frame kernels only (A/B timing), 1 = by batch size (default),
  // 2 = half-wave kernel for every batch it supports (tests)
  int halfwave = 1;
  // afx_plan_desc.flags & AFX_PLAN_NO_SIDE_STREAM: the rhythm tracker's kernels are enqueued on the batch's own stream
  // instead of its side stream (per-kernel durations of a profile are then not inflated by overlap)
  bool side_stream = true;
};

struct afx_batch {
  afx_plan* plan = nullptr;
  uint32_t mask = 0;
  int pcm_dtype = afx::kPcmF32;   // afx::kPcm*: the ABI's AFX_PCM_F32 / AFX_PCM_F64, or kPcmScaledF32 behind the LoadSample front end
  int32_t n_bufs = 0;
  std::vector<int64_t> frame_offset;  // [n_bufs+1]
  std::vector<int32_t> buf_status;    // [n_bufs]
  int64_t total_frames = 0;
  int n_chunks = 0;
  int grid_blocks = 0;
  afx::RecordLayout lay{};
  Workspace* ws = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  void* d_pcm = nullptr;
  afx::Chunk* d_chunks = nullptr;
  afx::ChunkRemaining* d_rem = nullptr;
  int32_t* d_chunk_first = nullptr;      // whitening kernels: [n_bufs + 1] into d_wchunks
  afx::Chunk* d_wchunks = nullptr;       // whitening kernels' chunk table
  int n_wchunks = 0;
  bool need_follow = false;
  afx::BufSpan* d_spans = nullptr;
  int32_t* d_efflen = nullptr;
  double* d_follower = nullptr;
  int chunk_frames = 0;
  double* d_rec = nullptr;
  double* d_mag = nullptr;
  int64_t* d_frame_offset = nullptr;
  double* d_stats = nullptr;
  unsigned* d_queue = nullptr;   // work-queue counter of the half-wave frame kernel (lives in the workspace)
  double* d_stat_tmp = nullptr;  // half-wave statistics class: raw sums per frame
  std::vector<int64_t> arena_off, used;  // per buffer: start and length (samples) of its analysed prefix in d_pcm
  std::vector<double> buf_scale;         // kPcmScaledF32: FinalScaling per buffer (empty otherwise)
  // rhythm tracker (AFX_D_RHYTHM)
  std::vector<int64_t> rt_offset;          // [n_bufs+1]: rows of the 512/128 frames
  std::vector<afx::RhythmFile> rt_files;   // [n_bufs]
  std::vector<int64_t> file_samples;       // [n_bufs]: mOriginalNumberOfSamples default (the buffer's / file's own length)
  std::vector<int32_t> file_offset;        // [n_bufs]: mDataOffset default
  std::vector<int32_t> file_rate;          // [n_bufs]: mOriginalSampleRate default (0: the plan's rate)
  bool rt_files_dirty = false;
  afx::RhythmFile* d_rt_files = nullptr;
  // long files of a small batch (afx_rhythm.hip): [n_long] file indices, [n_long + 1] round offsets (int32), [n_long + 1]
  // frame offsets (int64) in one device buffer; the polar scratch
  int32_t rt_n_long = 0, rt_long_rounds = 0;
  int64_t rt_long_rows = 0;
  void* d_rt_long = nullptr;
  float2* d_rt_polar = nullptr;
  float* d_rt_odf = nullptr;
  double* d_rt_onsets = nullptr;
  double* d_rt_scratch = nullptr;
  double* d_rt_scalars = nullptr;
  double* d_rt_stats = nullptr;
  int64_t* d_rt_foff = nullptr;
  // host copies of the tables that are uploaded asynchronously: they live as long as the batch, so creation does not
  // have to wait for the uploads (afx_batch_destroy synchronises the stream before they go)
  std::vector<afx::Chunk> h_chunks;
  std::vector<afx::ChunkRemaining> h_remaining;
  std::vector<int32_t> h_chunk_first;
  std::vector<afx::Chunk> h_wchunks;
  std::vector<afx::BufSpan> h_spans;
  std::vector<afx::LoadPlace> h_place;
  bool mag_wanted = false;
  bool ran = false;        // afx_batch_run has been enqueued at least once: the fetches have something to fetch
  bool halfwave = false;   // frames by the half-wave kernel: a wave walks two chunks at a time
};

namespace {

template <typename T>
int upload_tables_typed(afx_plan* p) {
  using C = cpx<T>;
  const int fft = p->desc.fft_size;
  std::vector<C> win(1024), t1(64), t2(1024), post(1024);
  for (int r = 0; r < 16; ++r)
    for (int lane = 0; lane < 64; ++lane) {
      const int n = 64 * r + lane;
      // 1/fft: kDivFwdByN (Fourier.cpp:265-270); 1/2: even/odd untangle of the half-size FFT
      win[n] = {(T)(p->window[2 * n] / (2.0 * fft)), (T)(p->window[2 * n + 1] / (2.0 * fft))};
      post[n] = twiddle<T>(lane + 64 * r, 2048);
    }
  // T1[jh][m2][jl] = w64^(m2 (4 jh + jl)): after the register transpose the lane row is jh and
  // the register is 4 m2 + jl
  for (int jh = 0; jh < 4; ++jh)
    for (int m2 = 0; m2 < 4; ++m2)
      for (int jl = 0; jl < 4; ++jl) t1[16 * jh + 4 * m2 + jl] = twiddle<T>((long long)m2 * (4 * jh + jl), 64);
  // T2[4 j2 + jl][16 jh + n2] = w1024^(n2 (4 jh + jl + 16 j2))
  for (int g = 0; g < 16; ++g)
    for (int lane = 0; lane < 64; ++lane) {
      const int j2 = g >> 2, jl = g & 3, jh = lane >> 4, n2 = lane & 15;
      t2[64 * g + lane] = twiddle<T>((long long)n2 * (4 * jh + jl + 16 * j2), 1024);
    }
  auto up = [](void** dst, const void* src, size_t bytes) -> hipError_t {
    hipError_t e = hipMalloc(dst, bytes);
    if (e != hipSuccess) return e;
    return hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
  };
  HIP_TRY(up(&p->dev.win, win.data(), win.size() * sizeof(C)));
  HIP_TRY(up(&p->dev.t1, t1.data(), t1.size() * sizeof(C)));
  HIP_TRY(up(&p->dev.t2, t2.data(), t2.size() * sizeof(C)));
  HIP_TRY(up(&p->dev.post, post.data(), post.size() * sizeof(C)));
  if (sizeof(T) == 8) {
    p->dev.t1_f64 = p->dev.t1; p->dev.t2_f64 = p->dev.t2; p->dev.post_f64 = p->dev.post;
  }
  return AFX_OK;
}

// tables of the half-wave kernels (afx_frames32.hip): bin / sample index = q + 32 row, always double
int upload_halfwave_tables(afx_plan* p) {
  using C = cpx<double>;
  const int fft = p->desc.fft_size;
  std::vector<C> win(1024), tw(1024), post(1024);
  for (int r = 0; r < 32; ++r)
    for (int q = 0; q < 32; ++q) {
      const int n = q + 32 * r;
      // 1/fft: kDivFwdByN (Fourier.cpp:265-270); 1/2: even/odd untangle of the half-size FFT; another 1/2: the
      // magnitude's Newton step returns twice the square root (mag_sqrt_mel, afx_frames32.hip)
      win[32 * r + q] = {p->window[2 * n] / (4.0 * fft), p->window[2 * n + 1] / (4.0 * fft)};
      tw[32 * r + q] = twiddle<double>((long long)r * q, 1024);   // [n2 = r][k1 = q]
      post[32 * r + q] = twiddle<double>(n, 2048);
    }
  std::vector<double> melw((size_t)afx::kMel32Pairs * 32, 0.0);
  int idx = 0;
  for (int r = 0; r < afx::kMel32Rows; ++r)
    for (int f = 0; f < afx::kNumCep; ++f)
      if (afx::mel32_touches(f, r)) {
        for (int q = 0; q < 32; ++q) melw[(size_t)idx * 32 + q] = p->mel[(size_t)f * afx::kHalf + 32 * r + q];
        ++idx;
      }
  auto up = [](void** dst, const void* src, size_t bytes) -> hipError_t {
    hipError_t e = hipMalloc(dst, bytes);
    if (e != hipSuccess) return e;
    return hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
  };
  HIP_TRY(up(&p->dev.win32, win.data(), win.size() * sizeof(C)));
  HIP_TRY(up(&p->dev.tw32, tw.data(), tw.size() * sizeof(C)));
  HIP_TRY(up(&p->dev.post32, post.data(), post.size() * sizeof(C)));
  HIP_TRY(up(&p->dev.melw32, melw.data(), melw.size() * sizeof(double)));
  return AFX_OK;
}

// tables of the rhythm tracker (afx_rhythm.hip); transcendental values come from the host's libm like the reference's
int upload_rhythm_tables(afx_plan* p) {
  using C = cpx<double>;
  const int n = 512;
  std::vector<double> win((size_t)n);
  const double delta = 1.0 / (double)(n - 1);   // TFftWindow::SFillBuffer, kHanning (Fourier.cpp:505, 545-551)
  for (int i = 0; i < n; ++i) win[(size_t)i] = 0.5 * (0.5 * (1.0 - std::cos(6.2831853071795864769252867665590 * (double)i * delta)));
  std::vector<C> tw(256), ut(256);
  for (int n2 = 0; n2 < 16; ++n2)
    for (int k1 = 0; k1 < 16; ++k1) tw[(size_t)n2 * 16 + k1] = twiddle<double>((long long)n2 * k1, 256);
  for (int r = 0; r < 16; ++r)
    for (int q = 0; q < 16; ++q) ut[(size_t)r * 16 + q] = twiddle<double>(q + 16 * r, 512);
  std::vector<double> canny(25);                // TCannyWindow(12, 16.0)::WindowValue, CannyWindow.cpp:72-78
  const double sq = 16.0 * 16.0;
  for (int i = -12; i <= 12; ++i) canny[(size_t)(i + 12)] = (double)i / sq * std::exp(-1.0 * (i * i) / (2.0 * sq));
  std::vector<double> ray((size_t)afx::kRayleighTable);   // rwv, beattracking.c:65, 105-108
  const double rayparam = 60. * p->desc.sample_rate / 120. / 128;
  for (int i = 0; i < afx::kRayleighTable; ++i)
    ray[(size_t)i] = ((double)(i + 1.) / (rayparam * rayparam)) * std::exp((-((i + 1.) * (i + 1.)) / (2. * (rayparam * rayparam))));
  auto up = [](double** dst, const void* src, size_t bytes) -> hipError_t {
    hipError_t e = hipMalloc((void**)dst, bytes);
    if (e != hipSuccess) return e;
    return hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
  };
  HIP_TRY(up(&p->dev.rt_window, win.data(), win.size() * sizeof(double)));
  HIP_TRY(up(&p->dev.rt_tw, tw.data(), tw.size() * sizeof(C)));
  HIP_TRY(up(&p->dev.rt_ut, ut.data(), ut.size() * sizeof(C)));
  HIP_TRY(up(&p->dev.rt_canny, canny.data(), canny.size() * sizeof(double)));
  HIP_TRY(up(&p->dev.rt_rayleigh, ray.data(), ray.size() * sizeof(double)));
  return AFX_OK;
}

int upload_tables(afx_plan* p) {
  int st = upload_tables_typed<double>(p);
  if (st != AFX_OK) return st;
  if ((st = upload_halfwave_tables(p)) != AFX_OK) return st;
  if ((st = upload_rhythm_tables(p)) != AFX_OK) return st;
  // packed mel rows: one 64-lane row per (r, f) pair the static cover lists, in the kernel's precision
  std::vector<double> melw((size_t)afx::kMelPairs * 64, 0.0);
  int idx = 0;
  for (int r = 0; r < afx::kMelRows; ++r)
    for (int f = 0; f < afx::kNumCep; ++f)
      if (afx::mel_touches(f, r)) {
        for (int lane = 0; lane < 64; ++lane)
          melw[(size_t)idx * 64 + lane] = p->mel[(size_t)f * afx::kHalf + 64 * r + lane];
        ++idx;
      }
  HIP_TRY(hipMalloc(&p->dev.melw, melw.size() * sizeof(double)));
  HIP_TRY(hipMemcpy(p->dev.melw, melw.data(), melw.size() * sizeof(double), hipMemcpyHostToDevice));
  // DCT-II basis exactly as xtract_dct evaluates it (vector.c:381-385)
  std::vector<double> dct(14 * 16, 0.0);
  for (int n = 0; n < 14; ++n)
    for (int m = 1; m <= 14; ++m) dct[16 * n + (m - 1)] = std::cos(M_PI * (n / (double)14) * (m - 0.5));
  HIP_TRY(hipMalloc((void**)&p->dev.dct, dct.size() * sizeof(double)));
  HIP_TRY(hipMemcpy(p->dev.dct, dct.data(), dct.size() * sizeof(double), hipMemcpyHostToDevice));
  return AFX_OK;
}

void free_tables(afx_plan* p) {
  if (p->dev.t1_f64 != p->dev.t1) { hipFree(p->dev.t1_f64); hipFree(p->dev.t2_f64); hipFree(p->dev.post_f64); }
  hipFree(p->dev.win); hipFree(p->dev.t1); hipFree(p->dev.t2); hipFree(p->dev.post);
  hipFree(p->dev.melw); hipFree(p->dev.dct);
  hipFree(p->dev.win32); hipFree(p->dev.tw32); hipFree(p->dev.post32); hipFree(p->dev.melw32);
  hipFree(p->dev.rt_window); hipFree(p->dev.rt_tw); hipFree(p->dev.rt_ut); hipFree(p->dev.rt_canny); hipFree(p->dev.rt_rayleigh);
  hipFree(p->dev.rs_filter);
  p->dev = DeviceTables{};
}

void ws_free(Workspace* w) {
  if (!w) return;
  for (Workspace::Buf* b : {&w->pcm, &w->chunks, &w->wchunks, &w->rem, &w->rec, &w->mag, &w->foff, &w->stats, &w->cfirst, &w->follower, &w->spans,
                            &w->efflen, &w->raw, &w->files, &w->scan, &w->partial, &w->place, &w->queue, &w->rt_files, &w->rt_long, &w->rt_polar, &w->rt_odf,
                            &w->rt_onsets, &w->rt_scratch, &w->rt_scalars, &w->rt_stats, &w->rt_foff, &w->stat_tmp, &w->rs_files, &w->rs_groups,
                            &w->rs_ngroups}) hipFree(b->p);
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

hipError_t ws_reserve(Workspace::Buf& b, size_t bytes) {
  if (bytes <= b.cap) return hipSuccess;
  hipFree(b.p);
  b.p = nullptr;
  b.cap = 0;
  const size_t want = bytes + bytes / 4 + 4096;
  hipError_t e = hipMalloc(&b.p, want);
  if (e != hipSuccess) return e;
  b.cap = want;
  return hipSuccess;
}

// Every per-frame series of the ABI: where it lives in afx_out / afx_stats_out / the device record, its
// width, and the mask bit that selects it.  The order is the record order.
struct FieldDesc {
  double* afx_out::*out;
  double* afx_stats_out::*stat;
  int32_t afx::RecordLayout::*off;
  int width;
  uint32_t bit;
};
#define AFX_FIELD(name, lay, width, bit) {&afx_out::name, &afx_stats_out::name, &afx::RecordLayout::lay, width, bit}
const FieldDesc kFields[] = {
    AFX_FIELD(mfcc, mfcc, 14, AFX_D_MFCC),
    AFX_FIELD(spectral_rms, srms, 1, AFX_D_SPECTRAL_RMS),
    AFX_FIELD(spectral_centroid, centroid, 1, AFX_D_SPECTRAL_CENTROID),
    AFX_FIELD(spectral_spread, spread, 1, AFX_D_SPECTRAL_SPREAD),
    AFX_FIELD(spectral_skewness, skew, 1, AFX_D_SPECTRAL_SKEWNESS),
    AFX_FIELD(spectral_kurtosis, kurt, 1, AFX_D_SPECTRAL_KURTOSIS),
    AFX_FIELD(spectral_rolloff, rolloff, 1, AFX_D_SPECTRAL_ROLLOFF),
    AFX_FIELD(spectral_flatness, flatness, 1, AFX_D_SPECTRAL_FLATNESS),
    AFX_FIELD(spectral_flux, flux, 1, AFX_D_SPECTRAL_FLUX),
    AFX_FIELD(spectrum_bands, bands, 28, AFX_D_SPECTRUM_BANDS),
    AFX_FIELD(amplitude_peak, amp_peak, 1, AFX_D_AMPLITUDE_PEAK),
    AFX_FIELD(amplitude_rms, amp_rms, 1, AFX_D_AMPLITUDE_RMS),
    AFX_FIELD(sub_rms, sub_rms, 14, AFX_D_BAND_FEATURES),
    AFX_FIELD(sub_flatness, sub_flat, 14, AFX_D_BAND_FEATURES),
    AFX_FIELD(sub_flux, sub_flux, 14, AFX_D_BAND_FEATURES),
    AFX_FIELD(sub_complexity, sub_cplx, 14, AFX_D_BAND_FEATURES),
    AFX_FIELD(sub_contrast, sub_contrast, 14, AFX_D_BAND_FEATURES),
    AFX_FIELD(spectral_contrast, contrast, 1, AFX_D_BAND_FEATURES),
    AFX_FIELD(amplitude_silence, silence, 1, AFX_D_AMPLITUDE_SILENCE),
    AFX_FIELD(amplitude_envelope, envelope, 1, AFX_D_AMPLITUDE_ENVELOPE),
    AFX_FIELD(spectral_complexity, complexity, 1, AFX_D_SPECTRAL_COMPLEXITY),
    AFX_FIELD(auto_correlation, autocorr, 1, AFX_D_AUTO_CORRELATION),
    AFX_FIELD(f0, f0, 1, AFX_D_F0),
    AFX_FIELD(f0_confidence, f0_conf, 1, AFX_D_F0),
    AFX_FIELD(failsafe_f0, f0_safe, 1, AFX_D_F0),
    AFX_FIELD(spectral_inharmonicity, inharm, 1, AFX_D_SPECTRAL_INHARMONICITY),
    AFX_FIELD(tristimulus1, tri1, 1, AFX_D_TRISTIMULUS),
    AFX_FIELD(tristimulus2, tri2, 1, AFX_D_TRISTIMULUS),
    AFX_FIELD(tristimulus3, tri3, 1, AFX_D_TRISTIMULUS),
};
#undef AFX_FIELD

afx::RecordLayout make_layout(uint32_t mask) {
  afx::RecordLayout l{};
  int off = 0;
  for (const FieldDesc& f : kFields) {
    l.*(f.off) = -1;
    if (mask & f.bit) { l.*(f.off) = off; off += f.width; }
  }
  l.stride = off;
  return l;
}

// which kernels a descriptor mask needs
constexpr uint32_t kSpectralBits = AFX_D_ALL_LOW_LEVEL | AFX_D_MAGNITUDE;
constexpr uint32_t kNeedsMagnitudes = AFX_D_MAGNITUDE | AFX_D_BAND_FEATURES | AFX_D_SPECTRAL_FLUX |
                                      AFX_D_SPECTRAL_COMPLEXITY | AFX_D_F0;
constexpr uint32_t kWhitenBits = AFX_D_SPECTRAL_COMPLEXITY | AFX_D_F0 | AFX_D_SPECTRAL_INHARMONICITY | AFX_D_TRISTIMULUS;
constexpr uint32_t kTimeBits = AFX_D_AMPLITUDE_SILENCE | AFX_D_AMPLITUDE_ENVELOPE | AFX_D_AUTO_CORRELATION | AFX_D_F0;
// the frame kernel sees the spectral bits; storing the magnitudes is its AFX_D_MAGNITUDE path
uint32_t frames_mask(uint32_t mask) {
  uint32_t m = mask & kSpectralBits;
  if (mask & kNeedsMagnitudes) m |= AFX_D_MAGNITUDE;
  // who reads bins above 768 of the stored magnitudes: the caller (AFX_D_MAGNITUDE) and the whitening kernels
  if (mask & (AFX_D_MAGNITUDE | AFX_D_SPECTRAL_COMPLEXITY | AFX_D_F0)) m |= afx::kFramesWholeSpectrum;
  // bands_kernel runs for this mask anyway and the spectral statistics are wanted: on the half-wave layout it takes them
  // from the stored magnitudes (seven sums more over rows it holds), the frame kernel only stores (magnitude class)
  if ((mask & 0xFEu) && (mask & (AFX_D_BAND_FEATURES | AFX_D_SPECTRAL_FLUX | AFX_D_SPECTRUM_BANDS))) m |= afx::kFramesStatsLater;
  return m;
}

// SampleDataAnalyzationLength, SampleAnalyser.cpp:760-764
int64_t analysed_length(const afx_plan* p, int64_t n_samples) {
  int64_t len = n_samples;
  if (p->desc.max_analysis_ms > 0) {
    const int64_t cap = ms_to_samples(p->desc.sample_rate, (float)p->desc.max_analysis_ms);
    len = std::min(len, cap);
  }
  return len;
}

int64_t num_frames(const afx_plan* p, int64_t n_samples) {
  const int64_t len = analysed_length(p, n_samples);
  if (len < p->desc.fft_size) return 0;
  return (len - p->desc.fft_size) / p->desc.hop_size + 1;  // SampleAnalyser.cpp:814
}

}  // namespace

extern "C" {

const char* afx_status_str(int status) {
  switch (status) {
    case AFX_OK: return "ok";
    case AFX_ERR_INVALID_ARG: return "invalid argument";
    case AFX_ERR_UNSUPPORTED: return "unsupported plan geometry";
    case AFX_ERR_NO_DEVICE: return "no HIP device";
    case AFX_ERR_OUT_OF_MEMORY: return "out of memory";
    case AFX_ERR_HIP: return "HIP runtime error";
    case AFX_ERR_BAD_BUFFER: return "bad buffer";
    default: return "unknown status";
  }
}

const char* afx_last_error(void) { return g_last_error.c_str(); }

// What this library was built with: the shipped library has no diagnostic or ablation switch set
// (tests/test_capi_cpu.py asserts it).
const char* afx_build_info(void) {
#if defined(AFX_STAMPS) && AFX_STAMPS
#define AFX_INFO_STAMPS "1"
#else
#define AFX_INFO_STAMPS "0"
#endif
#define AFX_INFO_ABL "0"   /* the ablation switches of round 1 are gone from the sources */
#ifndef AFX_SRC_HASH
#define AFX_SRC_HASH "unknown"
#endif
  return "afx abi=" "5" " arch=gfx950 stamps=" AFX_INFO_STAMPS " ablation=" AFX_INFO_ABL " src=" AFX_SRC_HASH;
}

int afx_plan_create(const afx_plan_desc* desc, afx_plan** out_plan) {
  if (!desc || !out_plan) return fail(AFX_ERR_INVALID_ARG, "null argument");
  *out_plan = nullptr;
  if (desc->sample_rate <= 0 || desc->fft_size <= 0 || desc->hop_size <= 0 ||
      (desc->fft_size & (desc->fft_size - 1)) || desc->max_analysis_ms < 0 ||
      (desc->precision != AFX_PRECISION_F64 && desc->precision != AFX_PRECISION_F32) ||
      desc->frame_kernel < AFX_FRAME_KERNEL_AUTO || desc->frame_kernel > AFX_FRAME_KERNEL_HALFWAVE ||
      (desc->flags & ~(int32_t)AFX_PLAN_NO_SIDE_STREAM))
    return fail(AFX_ERR_INVALID_ARG, "bad plan descriptor");
  if (desc->precision == AFX_PRECISION_F32)
    return fail(AFX_ERR_UNSUPPORTED,
                "AFX_PRECISION_F32 was removed: narrower than the reference's arithmetic (it missed the parity bar on tonal "
                "input) and no faster than the double path since the half-wave kernel");
  if (desc->sample_rate != afx::kSampleRate || desc->fft_size != afx::kFft || desc->hop_size != afx::kHop)
    return fail(AFX_ERR_UNSUPPORTED,
                "the HIP kernels are specialised for 44100 Hz / 2048 / 1024 (Crawler.cpp:41-43)");
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
    return fail(AFX_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU path)");
  if (desc->device < 0 || desc->device >= n_dev) return fail(AFX_ERR_NO_DEVICE, "device ordinal out of range");

  afx_plan* p = new (std::nothrow) afx_plan();
  if (!p) return fail(AFX_ERR_OUT_OF_MEMORY, "host allocation failed");
  p->desc = *desc;
  // SampleAnalyser.cpp:171-175: (int / int) stored in a double
  const double fpb = (double)(desc->sample_rate / desc->fft_size);
  p->first_bin = d2i_round(20.0 / fpb);
  p->last_bin = d2i_round(15500.0 / fpb);
  p->bin_count = p->last_bin - p->first_bin + 1;
  p->window = build_window(desc->fft_size);
  p->mel = build_mel(desc->fft_size / 2, (double)(desc->sample_rate / 2), 20.0, 15500.0, afx::kNumCep);

  // the kernels' static structure must cover the tables just built
  bool ok = (p->first_bin == afx::kFirstBin && p->last_bin == afx::kLastBin);
  for (int f = 0; f < afx::kNumCep && ok; ++f)
    for (int k = 0; k < afx::kHalf; ++k)
      if (p->mel[(size_t)f * afx::kHalf + k] != 0.0 && (k < afx::kMelLo[f] || k > afx::kMelHi[f])) ok = false;
  {
    // sub-band bin counts exactly as SampleAnalyser.cpp:2087-2100 derives them
    static const double sub_edges[afx::kNumSub] = {50.0, 100.0, 200.0, 400.0, 630.0, 920.0, 1270.0, 1720.0,
                                                   2320.0, 3150.0, 4400.0, 6400.0, 9500.0, 15500.0};
    const int first = d2i_round(20.0 / fpb);
    for (int b = 0; b < afx::kNumSub && ok; ++b) {
      const int start = (b == 0) ? first : d2i_round(sub_edges[b - 1] / fpb);
      const int end = d2i_round(sub_edges[b] / fpb);
      if (end - start + 1 != afx::kSubN[b]) ok = false;
    }
  }
  if (!ok) {
    delete p;
    return fail(AFX_ERR_UNSUPPORTED, "mel table / bin ranges outside the kernels' static cover");
  }

  hipError_t e = hipSetDevice(desc->device);
  if (e != hipSuccess) { delete p; return hip_fail(e, "hipSetDevice"); }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, desc->device) == hipSuccess) p->cu_count = prop.multiProcessorCount;
  p->halfwave = (desc->frame_kernel == AFX_FRAME_KERNEL_WAVE64) ? 0 : (desc->frame_kernel == AFX_FRAME_KERNEL_HALFWAVE ? 2 : 1);
  p->side_stream = !(desc->flags & AFX_PLAN_NO_SIDE_STREAM);
  const int st = upload_tables(p);
  if (st != AFX_OK) { free_tables(p); delete p; return st; }
  if (hipStreamCreateWithFlags(&p->up_stream, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&p->down_stream, hipStreamNonBlocking) != hipSuccess) {
    if (p->up_stream) hipStreamDestroy(p->up_stream);
    free_tables(p);
    delete p;
    return fail(AFX_ERR_HIP, "hipStreamCreate(copy streams)");
  }
  *out_plan = p;
  return AFX_OK;
}

static void plan_release(afx_plan* plan) {
  if (plan->refs.fetch_sub(1) != 1) return;
  hipSetDevice(plan->desc.device);
  for (Workspace* w : plan->pool) ws_free(w);
  plan->pool.clear();
  if (plan->up_stream) hipStreamDestroy(plan->up_stream);
  if (plan->down_stream) hipStreamDestroy(plan->down_stream);
  free_tables(plan);
  delete plan;
}

void afx_plan_destroy(afx_plan* plan) {
  if (!plan) return;
  plan_release(plan);   // deferred until the last batch of this plan is destroyed
}

int afx_plan_get_window(const afx_plan* plan, double* out) {
  if (!plan || !out) return fail(AFX_ERR_INVALID_ARG, "null argument");
  std::memcpy(out, plan->window.data(), plan->window.size() * sizeof(double));
  return AFX_OK;
}
int afx_plan_get_mel_table(const afx_plan* plan, double* out) {
  if (!plan || !out) return fail(AFX_ERR_INVALID_ARG, "null argument");
  std::memcpy(out, plan->mel.data(), plan->mel.size() * sizeof(double));
  return AFX_OK;
}
int afx_plan_get_bin_range(const afx_plan* plan, int32_t* first_bin, int32_t* bin_count) {
  if (!plan) return fail(AFX_ERR_INVALID_ARG, "null argument");
  if (first_bin) *first_bin = plan->first_bin;
  if (bin_count) *bin_count = plan->bin_count;
  return AFX_OK;
}

int64_t afx_num_frames(const afx_plan* plan, int64_t n_samples) {
  if (!plan || n_samples < 0) return 0;
  return num_frames(plan, n_samples);
}

int64_t afx_algorithmic_bytes_per_frame(const afx_plan* plan, uint32_t mask, int32_t pcm_dtype) {
  if (!plan) return 0;
  const int64_t in = (int64_t)plan->desc.hop_size * (pcm_dtype == AFX_PCM_F64 ? 8 : 4);
  int64_t out = (int64_t)make_layout(mask).stride * 8;
  if (mask & AFX_D_MAGNITUDE) out += (int64_t)afx::kHalf * 8;
  return in + out;
}

}  // extern "C"

namespace {

// SampleDurationInSeconds / OnsetOffsetInSeconds of every buffer (SampleAnalyser.cpp:1001-1004): TAudioMath::SamplesToMs
// is float arithmetic (AudioMath.inl:134-137) and the division by the int 1000 stays in float
void set_rhythm_context(afx_batch* b, const afx_file_info* info) {
  for (int32_t i = 0; i < b->n_bufs; ++i) {
    const int own_rate = (i < (int32_t)b->file_rate.size() && b->file_rate[(size_t)i] > 0) ? b->file_rate[(size_t)i] : b->plan->desc.sample_rate;
    const int rate = (info && info[i].original_sample_rate > 0) ? info[i].original_sample_rate : own_rate;
    const int samples = (int)(info ? info[i].original_samples : b->file_samples[(size_t)i]);
    const int offset = info ? info[i].data_offset : b->file_offset[(size_t)i];
    afx::RhythmFile& rf = b->rt_files[(size_t)i];
    rf.duration_s = (double)(((float)samples / ((float)rate / 1000.0f)) / 1000);
    rf.offset_s = (double)(((float)offset / ((float)rate / 1000.0f)) / 1000);
  }
  b->rt_files_dirty = true;
}

// Common tail of batch creation: lengths[i] = samples of buffer i as AnalyzeLowLevelDescriptors would
// see them; fill() puts the analysed prefix of every buffer at arena_off[i] of b->d_pcm (element
// size esz) using b->stream.
template <typename Fill>
int build_batch(afx_plan* plan, int32_t n_bufs, uint32_t mask, int dtype, const std::vector<int64_t>& lengths,
                const std::vector<int32_t>& status, bool zero_arena, Fill fill, afx_batch** out_batch,
                Workspace* acquired = nullptr, const std::vector<int64_t>* file_samples = nullptr,
                const std::vector<int32_t>* file_offset = nullptr, bool wait_for_uploads = true,
                const std::vector<int32_t>* file_rate = nullptr, const std::vector<double>* scales = nullptr) {
  const bool want_stats = (mask & AFX_D_STATISTICS) != 0;
  mask &= ~(uint32_t)AFX_D_STATISTICS;   // the kernels see the descriptor bits only

  afx_batch* b = new (std::nothrow) afx_batch();
  if (!b) {
    ws_release(plan, acquired);
    return fail(AFX_ERR_OUT_OF_MEMORY, "host allocation failed");
  }
  b->plan = plan;
  plan->refs.fetch_add(1);
  b->mask = mask;
  b->n_bufs = n_bufs;
  b->lay = make_layout(mask);
  // flux, the sub-band descriptors and the whitened-spectrum neighbours are computed from the stored
  // magnitudes by later kernels
  b->mag_wanted = (mask & kNeedsMagnitudes) != 0;
  const uint32_t fmask = frames_mask(mask);
  b->frame_offset.assign((size_t)n_bufs + 1, 0);
  b->buf_status = status;
  b->pcm_dtype = dtype;
  const size_t esz = (dtype == afx::kPcmF64) ? 8 : 4;
  // kPcmScaledF32 (the LoadSample front end): the arena holds the float mono signal, the buffer's FinalScaling rides
  // along in the chunk table / buffer spans / rhythm file records
  auto scale_of = [&](int i) { return scales ? (*scales)[(size_t)i] : 1.0; };
  if (scales) b->buf_scale = *scales;

  // arena offsets: every buffer starts on a 16-byte boundary and only the analysed prefix
  // (SampleAnalyser.cpp:760-764) of buffers that yield at least one frame is kept
  b->arena_off.assign((size_t)n_bufs, 0);
  b->used.assign((size_t)n_bufs, 0);
  int64_t arena = 0, frames = 0;
  for (int i = 0; i < n_bufs; ++i) {
    b->frame_offset[i] = frames;
    if (b->buf_status[i] != AFX_OK) continue;
    const int64_t f = num_frames(plan, lengths[i]);
    int64_t keep = 0;
    if (f > 0) {
      keep = (f - 1) * plan->desc.hop_size + plan->desc.fft_size;
      // the second rising-slope search of CalcAutoCorrelation (SampleAnalyser.cpp:2343-2356) may look up to
      // 33 samples past the last frame when the buffer has them
      if (mask & AFX_D_AUTO_CORRELATION) keep = std::min<int64_t>(lengths[i], keep + 64);
      frames += f;
    }
    // the rhythm tracker's 512/128 frames reach up to the end of the analysed prefix (SampleAnalyser.cpp:991)
    if (mask & AFX_D_RHYTHM) keep = std::max(keep, analysed_length(plan, lengths[i]));
    // CalcEffectiveLength scans the whole buffer, also beyond the analysed 20 s (SampleAnalyser.cpp:754)
    if (mask & AFX_D_EFFECTIVE_LENGTH) keep = lengths[i];
    if (keep > 0) {
      b->used[i] = keep;
      b->arena_off[i] = arena;
      arena += (keep + 3) & ~(int64_t)3;
    }
  }
  b->frame_offset[n_bufs] = frames;
  b->total_frames = frames;
  if (frames > 0x7FFFFF00LL) {
    delete b;
    plan->refs.fetch_sub(1);
    ws_release(plan, acquired);
    return fail(AFX_ERR_INVALID_ARG, "more than 2^31 frames in one batch");
  }

  // chunking: K consecutive frames per wave; enough chunks to fill the chip, long enough to
  // amortise the 2048-sample lead-in of each chunk
  // K is picked to minimise (rounds of the wave slots) x (frames per chunk + lead-in): long chunks for
  // big batches, one round of short chunks when the batch barely fills the chip
  // the half-wave kernel pays a longer prologue per chunk: it serves batches that give every half-wave slot
  // several frames; smaller ones (one short file per call) stay with the 64-lane kernel
  b->halfwave = plan->halfwave && afx::frames_use_halfwave(fmask, plan->desc.precision, dtype) &&
                (plan->halfwave == 2 || frames >= 8 * (int64_t)plan->cu_count * afx::frames32_waves_per_block() * 2);
  // the half-wave full class always leaves the magnitudes (bands_kernel takes flux, the 28 bands and the sub-band
  // descriptors from them)
  if (b->halfwave && afx::frames32_class(fmask) >= 2) b->mag_wanted = true;
  const int waves_per_block = b->halfwave ? afx::frames32_waves_per_block() : afx::frames_waves_per_block(fmask);
  const int64_t slots = (int64_t)plan->cu_count * waves_per_block * (b->halfwave ? 2 : 1);
  // what a chunk costs before its first frame, in frames: the 64-lane kernel loads one hop; the half-wave kernel loads
  // the overlap rows and the window table and starts its hop DMA (measured on the C4 share, profiles/r04)
  constexpr double kHalfwavePrologue = 0.5;
  int K = 32;
  {
    double best = 1e300;
    // (acorr_kernel transforms frames two at a time, frames 2 i and 2 i + 1 of a buffer as the real and imaginary part of
    // one complex sequence: chunks of an even number of frames keep a frame's partner -- and with it the rounding of
    // its result -- independent of how the batch was cut)
    const int k_step = (mask & AFX_D_AUTO_CORRELATION) ? 2 : 1;
    for (int k = k_step; k <= 32; k += k_step) {
      int64_t nchunks = 0;
      for (int i = 0; i < n_bufs; ++i) nchunks += (b->frame_offset[i + 1] - b->frame_offset[i] + k - 1) / k;
#if defined(AFX_X_TUNE)   // timing experiments only (never the shipped library)
      static const double x_prologue = std::getenv("AFX_X_PROLOGUE") ? std::atof(std::getenv("AFX_X_PROLOGUE")) : -1.0;
      const double prologue = x_prologue >= 0.0 ? x_prologue : (b->halfwave ? kHalfwavePrologue : 0.5);
#else
      const double prologue = b->halfwave ? kHalfwavePrologue : 0.5;
#endif
      // The 64-lane frame kernel gives every wave the same number of chunks: rounds of the wave slots x (frames per chunk
      // + lead-in).  Every kernel of the half-wave batches draws its chunks from a work queue: the slots share the work
      // (frames + lead-ins) evenly and run dry within a fraction of a chunk of each other.
      // (the 0.4: measured -- C3, 1 000 files of 82 frames: K = 6, 39.3 M frames/s against 35.8 M with the K = 22 of the
      // rounds model; the headline batch K = 32, 499 against 495 M with K = 25; the C4 share K = 10, 41.6 against 42.3 M
      // with K = 8; tools/gpu_r04_k.sh)
#ifndef AFX_X_TAIL
#define AFX_X_TAIL 0.4
#endif
      const double cost = b->halfwave ? ((double)frames + (double)nchunks * prologue) / (double)slots + AFX_X_TAIL * k
                                      : (double)((nchunks + slots - 1) / slots) * (k + prologue);
      if (cost < best - 1e-9 || (std::fabs(cost - best) <= 1e-9 && k > K)) { best = cost; K = k; }
    }
  }
  std::vector<afx::Chunk>& chunks = b->h_chunks;
  std::vector<afx::ChunkRemaining>& remaining = b->h_remaining;
  std::vector<int32_t>& chunk_first = b->h_chunk_first;   // (of the whitening kernels' table, below)
  b->chunk_frames = K;
  // When the half-wave frame kernel is the only consumer of the chunk table (it draws chunks from a work queue),
  // the last part of every buffer is cut into short chunks and the table is ordered long chunks first: the waves
  // run dry within a quarter of a long chunk's time of each other instead of a whole one.
  const bool guided = b->halfwave && K >= 8 && !(mask & (kTimeBits | kWhitenBits | AFX_D_BAND_FEATURES | AFX_D_SPECTRAL_FLUX));
  const int Ks = guided ? K / 4 : K;
  for (int i = 0; i < n_bufs; ++i) {
    const int64_t f = b->frame_offset[i + 1] - b->frame_offset[i];
    const int64_t f_long = guided ? (f * 7 / 8) / K * K : f;   // frames covered by chunks of K
    for (int64_t f0 = 0; f0 < f;) {
      const int k = (f0 < f_long) ? K : Ks;
      remaining.push_back((afx::ChunkRemaining)std::min<int64_t>(lengths[i] - f0 * plan->desc.hop_size, 1 << 30));
      afx::Chunk c;
      const bool first = (f0 == 0);
      c.sample_off = b->arena_off[i] + f0 * plan->desc.hop_size;
      c.frame0 = (int32_t)(b->frame_offset[i] + f0);
      c.nframes = (int16_t)std::min<int64_t>(k, f - f0);
      c.flags = (int16_t)(first ? afx::kChunkFirstOfBuffer : 0);
      c.scale = scale_of(i);
      chunks.push_back(c);
      f0 += k;
    }
  }
  if (guided)
    std::stable_sort(chunks.begin(), chunks.end(), [](const afx::Chunk& x, const afx::Chunk& y) { return x.nframes > y.nframes; });
  b->n_chunks = (int)chunks.size();
  // one workgroup per CU (its LDS holds the shared tables plus one exchange plane per wave);
  // waves walk the chunk list with a grid stride
  const int64_t wave_items = b->halfwave ? (b->n_chunks + 1) / 2 : b->n_chunks;
  b->grid_blocks = (int)std::min<int64_t>((wave_items + waves_per_block - 1) / waves_per_block,
                                          (int64_t)plan->cu_count);
  if (b->grid_blocks < 1) b->grid_blocks = 1;

  auto cleanup = [&](int st) { afx_batch_destroy(b); return st; };
  hipError_t e = hipSuccess;
  b->ws = acquired ? acquired : ws_acquire(plan, &e);
  if (!b->ws) return cleanup(hip_fail(e, "workspace"));
  Workspace& w = *b->ws;
  b->stream = w.stream; b->ev0 = w.ev0; b->ev1 = w.ev1;
  if (arena > 0) {
    if ((e = ws_reserve(w.pcm, (size_t)arena * esz + 64)) != hipSuccess) return cleanup(hip_fail(e, "hipMalloc(pcm)"));
    b->d_pcm = w.pcm.p;
    if (zero_arena && (e = hipMemsetAsync(b->d_pcm, 0, (size_t)arena * esz, b->stream)) != hipSuccess) return cleanup(hip_fail(e, "hipMemset"));
    const int st = fill(b);
    if (st != AFX_OK) return cleanup(st);
  }
  if (b->n_chunks > 0) {
    if ((e = ws_reserve(w.chunks, chunks.size() * sizeof(afx::Chunk))) != hipSuccess) return cleanup(hip_fail(e, "hipMalloc(chunks)"));
    b->d_chunks = (afx::Chunk*)w.chunks.p;
    if ((e = hipMemcpyAsync(b->d_chunks, chunks.data(), chunks.size() * sizeof(afx::Chunk), hipMemcpyHostToDevice, b->stream)) != hipSuccess) return cleanup(hip_fail(e, "hipMemcpy(chunks)"));
  }
  if (b->n_chunks > 0) {
    if (!w.queue.p) {
      // zeroed once: every launch advances its counter by its number of items (afx_internal.h: WorkQueue)
      if ((e = ws_reserve(w.queue, afx::kQueueSlots * sizeof(unsigned))) != hipSuccess) return cleanup(hip_fail(e, "hipMalloc(queue)"));
      if ((e = hipMemsetAsync(w.queue.p, 0, afx::kQueueSlots * sizeof(unsigned), b->stream)) != hipSuccess) return cleanup(hip_fail(e, "hipMemset(queue)"));
      for (unsigned& c : w.queue_count) c = 0;
    }
    b->d_queue = (unsigned*)w.queue.p;
  }
  if (b->n_chunks > 0 && b->halfwave) {
    if (fmask != 1u) {   // statistics class: raw sums per frame for its closed-form kernel
      if ((e = ws_reserve(w.stat_tmp, (size_t)frames * afx::frames32_stat_tmp_doubles() * sizeof(double))) != hipSuccess) return cleanup(hip_fail(e, "hipMalloc(stat_tmp)"));
      b->d_stat_tmp = (double*)w.stat_tmp.p;
    }
  }
  if (b->n_chunks > 0 && (mask & kTimeBits)) {
    if ((e = ws_reserve(w.rem, remaining.size() * sizeof(afx::ChunkRemaining))) != hipSuccess) return cleanup(hip_fail(e, "hipMalloc(remaining)"));
    b->d_rem = (afx::ChunkRemaining*)w.rem.p;
    if ((e = hipMemcpyAsync(b->d_rem, remaining.data(), remaining.size() * sizeof(afx::ChunkRemaining), hipMemcpyHostToDevice, b->stream)) != hipSuccess) return cleanup(hip_fail(e, "hipMemcpy(remaining)"));
  }
  if (b->n_chunks > 0 && (mask & kWhitenBits)) {
    // The whitening kernels walk their own chunk table.  Their only state across frames is the follower, which starts
    // from the reset value at a buffer's first frame: in a batch of thousands of short files (a crawl's typical batch)
    // a chunk is the whole file -- every chunk then starts from the reset state and follow_kernel (one more pass over
    // the 8 KiB of magnitudes per frame) is not needed; long files, and batches too small to fill the chip with one
    // wave per file, keep chunks of K frames whose start states follow_kernel provides.
    constexpr int64_t kWholeFileFrames = 128;
    const bool whole_files = n_bufs >= 768;     // one wave per file on at least three quarters of the chip's 1 024 SIMDs
    std::vector<afx::Chunk>& wchunks = b->h_wchunks;
    chunk_first.assign((size_t)n_bufs + 1, 0);
    for (int i = 0; i < n_bufs; ++i) {
      const int64_t f = b->frame_offset[i + 1] - b->frame_offset[i];
      chunk_first[(size_t)i] = (int32_t)wchunks.size();
      const int64_t step = (whole_files && f <= kWholeFileFrames) ? std::max<int64_t>(f, 1) : K;
      if (f > step) b->need_follow = true;
      for (int64_t f0 = 0; f0 < f; f0 += step) {
        afx::Chunk c;
        c.sample_off = b->arena_off[i] + f0 * plan->desc.hop_size;
        c.frame0 = (int32_t)(b->frame_offset[i] + f0);
        c.nframes = (int16_t)std::min<int64_t>(step, f - f0);
        c.flags = (int16_t)(f0 == 0 ? afx::kChunkFirstOfBuffer : 0);
        c.scale = scale_of(i);
        wchunks.push_back(c);
      }
    }
    chunk_first[(size_t)n_bufs] = (int32_t)wchunks.size();
    b->n_wchunks = (int)wchunks.size();
    if ((e = ws_reserve(w.wchunks, wchunks.size() * sizeof(afx::Chunk))) != hipSuccess) return cleanup(hip_fail(e, "hipMalloc(whitening chunks)"));
    b->d_wchunks = (afx::Chunk*)w.wchunks.p;
    if ((e = hipMemcpyAsync(b->d_wchunks, wchunks.data(), wchunks.size() * sizeof(afx::Chunk), hipMemcpyHostToDevice, b->stream)) != hipSuccess) return cleanup(hip_fail(e, "hipMemcpy(whitening chunks)"));
    if ((e = ws_reserve(w.cfirst, chunk_first.size() * sizeof(int32_t))) != hipSuccess) return cleanup(hip_fail(e, "hipMalloc(chunk_first)"));
    b->d_chunk_first = (int32_t*)w.cfirst.p;
    if ((e = hipMemcpyAsync(b->d_chunk_first, chunk_first.data(), chunk_first.size() * sizeof(int32_t), hipMemcpyHostToDevice, b->stream)) != hipSuccess) return cleanup(hip_fail(e, "hipMemcpy(chunk_first)"));
    if ((mask & AFX_D_SPECTRAL_COMPLEXITY) && b->need_follow) {
      if ((e = ws_reserve(w.follower, (size_t)b->n_wchunks * afx::kHalf * sizeof(double))) != hipSuccess) return cleanup(hip_fail(e, "hipMalloc(follower)"));
      b->d_follower = (double*)w.follower.p;
    }
  }
  if (n_bufs > 0 && (mask & AFX_D_EFFECTIVE_LENGTH)) {
    std::vector<afx::BufSpan>& spans = b->h_spans;
    spans.resize((size_t)n_bufs);
    for (int i = 0; i < n_bufs; ++i) spans[(size_t)i] = afx::BufSpan{b->arena_off[i], b->used[i], scale_of(i)};
    if ((e = ws_reserve(w.spans, spans.size() * sizeof(afx::BufSpan))) != hipSuccess) return cleanup(hip_fail(e, "hipMalloc(spans)"));
    b->d_spans = (afx::BufSpan*)w.spans.p;
    if ((e = hipMemcpyAsync(b->d_spans, spans.data(), spans.size() * sizeof(afx::BufSpan), hipMemcpyHostToDevice, b->stream)) != hipSuccess) return cleanup(hip_fail(e, "hipMemcpy(spans)"));
    if ((e = ws_reserve(w.efflen, (size_t)n_bufs * 6 * sizeof(int32_t))) != hipSuccess) return cleanup(hip_fail(e, "hipMalloc(efflen)"));
    b->d_efflen = (int32_t*)w.efflen.p;
  }
  if (frames > 0 && b->lay.stride > 0) {
    if ((e = ws_reserve(w.rec, (size_t)frames * b->lay.stride * sizeof(double))) != hipSuccess) return cleanup(hip_fail(e, "hipMalloc(rec)"));
    b->d_rec = (double*)w.rec.p;
  }
  if (frames > 0 && b->mag_wanted) {
    // (+ 1 row: the half-wave full class sends the stores of frames past a chunk's end there)
    if ((e = ws_reserve(w.mag, (size_t)(frames + 1) * afx::kHalf * sizeof(double))) != hipSuccess) return cleanup(hip_fail(e, "hipMalloc(mag)"));
    b->d_mag = (double*)w.mag.p;
  }
  if ((want_stats || (mask & kWhitenBits)) && n_bufs > 0 && b->lay.stride > 0) {
    if ((e = ws_reserve(w.foff, b->frame_offset.size() * sizeof(int64_t))) != hipSuccess) return cleanup(hip_fail(e, "hipMalloc(frame_offset)"));
    b->d_frame_offset = (int64_t*)w.foff.p;
    if ((e = hipMemcpyAsync(b->d_frame_offset, b->frame_offset.data(), b->frame_offset.size() * sizeof(int64_t), hipMemcpyHostToDevice, b->stream)) != hipSuccess) return cleanup(hip_fail(e, "hipMemcpy(frame_offset)"));
  }
  if (want_stats && n_bufs > 0 && b->lay.stride > 0) {
    if ((e = ws_reserve(w.stats, (size_t)n_bufs * b->lay.stride * 13 * sizeof(double))) != hipSuccess) return cleanup(hip_fail(e, "hipMalloc(stats)"));
    b->d_stats = (double*)w.stats.p;
  }
  if ((mask & AFX_D_RHYTHM) && n_bufs > 0) {
    // 512/128 frames of every buffer's analysed prefix: for (n = 0; n + 511 < length; n += 128), SampleAnalyser.cpp:991
    b->rt_offset.assign((size_t)n_bufs + 1, 0);
    b->rt_files.assign((size_t)n_bufs, afx::RhythmFile{});
    b->file_samples.assign((size_t)n_bufs, 0);
    b->file_offset.assign((size_t)n_bufs, 0);
    if (file_rate) b->file_rate = *file_rate;
    int64_t rows = 0;
    for (int i = 0; i < n_bufs; ++i) {
      b->rt_offset[(size_t)i] = rows;
      afx::RhythmFile& rf = b->rt_files[(size_t)i];
      rf.sample_off = b->arena_off[i];
      rf.scale = scale_of(i);
      rf.frame0 = rows;
      const int64_t len = (b->buf_status[i] == AFX_OK) ? std::min(analysed_length(plan, lengths[i]), b->used[i]) : 0;
      rf.frames = (len >= 512) ? (int32_t)((len - 512) / 128 + 1) : 0;
      rows += rf.frames;
      b->file_samples[(size_t)i] = file_samples ? (*file_samples)[(size_t)i] : lengths[i];
      b->file_offset[(size_t)i] = file_offset ? (*file_offset)[(size_t)i] : 0;
    }
    b->rt_offset[(size_t)n_bufs] = rows;
    // A batch of few files does not fill the chip with one workgroup per file, and a long file is hundreds of
    // dependent rounds for its workgroup (a lone 20 s file: 430 rounds, 5.7 ms): such files take the three-kernel path
    std::vector<int32_t> long_files, long_round_off;
    std::vector<int64_t> long_frame_off;
    if (n_bufs <= afx::kRhythmLongBatchFiles) {
      int32_t rounds = 0;
      int64_t lrows = 0;
      for (int i = 0; i < n_bufs; ++i)
        if (b->rt_files[(size_t)i].frames >= afx::kRhythmLongFrames) {
          long_files.push_back(i);
          long_round_off.push_back(rounds);
          long_frame_off.push_back(lrows);
          b->rt_files[(size_t)i].long_slot = (int32_t)long_files.size();
          rounds += (b->rt_files[(size_t)i].frames + 15) / 16;
          lrows += (b->rt_files[(size_t)i].frames + afx::kRhythmLongPad - 1) / afx::kRhythmLongPad * afx::kRhythmLongPad;
        }
      long_round_off.push_back(rounds);
      long_frame_off.push_back(lrows);
      b->rt_n_long = (int32_t)long_files.size();
      b->rt_long_rounds = rounds;
      if (b->rt_n_long > 0) {
        const size_t nl = long_files.size();
        // layout: int64 frame offsets [nl + 1], then int32 round offsets [nl + 1], then int32 file indices [nl]
        std::vector<unsigned char> blob((nl + 1) * 8 + (nl + 1) * 4 + nl * 4);
        std::memcpy(blob.data(), long_frame_off.data(), (nl + 1) * 8);
        std::memcpy(blob.data() + (nl + 1) * 8, long_round_off.data(), (nl + 1) * 4);
        std::memcpy(blob.data() + (nl + 1) * 12, long_files.data(), nl * 4);
        if ((e = ws_reserve(w.rt_long, blob.size())) != hipSuccess) return cleanup(hip_fail(e, "hipMalloc(rhythm long files)"));
        // (magnitude, phase) pairs, then the follower's float per bin: 12 bytes per bin and frame
        if ((e = ws_reserve(w.rt_polar, (size_t)lrows * 256 * (sizeof(float2) + sizeof(float)))) != hipSuccess) return cleanup(hip_fail(e, "hipMalloc(rhythm polar)"));
        b->rt_long_rows = lrows;
        b->d_rt_long = w.rt_long.p;
        b->d_rt_polar = (float2*)w.rt_polar.p;
        // (blocking copy: the blob is a local)
        if ((e = hipMemcpyAsync(b->d_rt_long, blob.data(), blob.size(), hipMemcpyHostToDevice, b->stream)) != hipSuccess) return cleanup(hip_fail(e, "hipMemcpy(rhythm long files)"));
        if ((e = hipStreamSynchronize(b->stream)) != hipSuccess) return cleanup(hip_fail(e, "hipStreamSynchronize"));
      }
    }
    set_rhythm_context(b, nullptr);
    if ((e = ws_reserve(w.rt_files, b->rt_files.size() * sizeof(afx::RhythmFile))) != hipSuccess) return cleanup(hip_fail(e, "hipMalloc(rhythm files)"));
    b->d_rt_files = (afx::RhythmFile*)w.rt_files.p;
    // (the stream is synchronised below, before the pageable source can change)
    if ((e = hipMemcpyAsync(b->d_rt_files, b->rt_files.data(), b->rt_files.size() * sizeof(afx::RhythmFile), hipMemcpyHostToDevice, b->stream)) != hipSuccess) return cleanup(hip_fail(e, "hipMemcpy(rhythm files)"));
    b->rt_files_dirty = false;
    if ((e = ws_reserve(w.rt_scalars, (size_t)n_bufs * AFX_NUM_RHYTHM_SCALARS * sizeof(double))) != hipSuccess) return cleanup(hip_fail(e, "hipMalloc(rhythm scalars)"));
    b->d_rt_scalars = (double*)w.rt_scalars.p;
    if (rows > 0) {
      if ((e = ws_reserve(w.rt_odf, (size_t)rows * 2 * sizeof(float))) != hipSuccess) return cleanup(hip_fail(e, "hipMalloc(onset functions)"));
      if ((e = ws_reserve(w.rt_onsets, (size_t)rows * 2 * sizeof(double))) != hipSuccess) return cleanup(hip_fail(e, "hipMalloc(onsets)"));
      if ((e = ws_reserve(w.rt_scratch, (size_t)rows * 8 * sizeof(double))) != hipSuccess) return cleanup(hip_fail(e, "hipMalloc(rhythm scratch)"));
      b->d_rt_odf = (float*)w.rt_odf.p; b->d_rt_onsets = (double*)w.rt_onsets.p; b->d_rt_scratch = (double*)w.rt_scratch.p;
    }
    if (want_stats) {
      if ((e = ws_reserve(w.rt_foff, b->rt_offset.size() * sizeof(int64_t))) != hipSuccess) return cleanup(hip_fail(e, "hipMalloc(rhythm offsets)"));
      b->d_rt_foff = (int64_t*)w.rt_foff.p;
      if ((e = hipMemcpyAsync(b->d_rt_foff, b->rt_offset.data(), b->rt_offset.size() * sizeof(int64_t), hipMemcpyHostToDevice, b->stream)) != hipSuccess) return cleanup(hip_fail(e, "hipMemcpy(rhythm offsets)"));
      if ((e = ws_reserve(w.rt_stats, (size_t)n_bufs * 2 * 13 * sizeof(double))) != hipSuccess) return cleanup(hip_fail(e, "hipMalloc(rhythm stats)"));
      b->d_rt_stats = (double*)w.rt_stats.p;
    }
  }
  // the caller's PCM buffers (afx_batch_create) may go away when this returns; the tables uploaded above are the batch's own
  if (wait_for_uploads && (e = hipStreamSynchronize(b->stream)) != hipSuccess) return cleanup(hip_fail(e, "hipStreamSynchronize"));
  *out_batch = b;
  return AFX_OK;
}

// AFX_TIMING=1 in the environment: wall time of the phases of afx_batch_create_from_raw, summed over calls and printed
// when the process ends (diagnostic for pipelines; three clock reads per call otherwise)
struct CreateTiming {
  std::atomic<long long> ns[4]{};   // upload + scan + wait, host placement, build + LoadSample write + wait, calls
  bool on = std::getenv("AFX_TIMING") != nullptr;
  ~CreateTiming() {
    if (on && ns[3].load())
      std::fprintf(stderr, "[afx timing] create_from_raw x%lld: upload + scan + wait %.1f ms, placement %.1f ms, build + write + wait %.1f ms\n",
                   ns[3].load(), ns[0].load() * 1e-6, ns[1].load() * 1e-6, ns[2].load() * 1e-6);
  }
};
CreateTiming g_create_timing;

inline long long now_ns() {
  return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// The host's waits.  hipStreamSynchronize / hipEventSynchronize spin (one busy CPU per waiting thread); with
// afx_plan_set_blocking_wait the thread polls the workspace's copy event and sleeps in between.  The runtime's own
// alternatives did not serve: hipEventBlockingSync alone changes nothing here (the eight workers of a crawl still keep
// 6.6 CPUs busy), and hipDeviceScheduleBlockingSync, set on a device that is already active, left a later
// hipStreamSynchronize hanging.
constexpr int kNapCeilingUs = 300;
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
struct Download { void* dst; const void* src; size_t bytes; };
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

// The converter's filter: right wing of a Kaiser-windowed sinc, Nmult = 35 zero crossings x 4096 values each
// (resample.c:104-124 -> lrsLpFilter / Izero, filterkit.c:66-113; roll-off 0.9, beta 6), computed in double and stored
// as float like the library does.  Uploaded once per plan, by the first batch that needs it.
hipError_t resample_filter_table(afx_plan* plan) {
  std::lock_guard<std::mutex> lock(plan->pool_mutex);
  if (plan->dev.rs_filter) return hipSuccess;
  constexpr int kNpc = 4096, kNwing = kNpc * (35 - 1) / 2;
  auto izero = [](double x) {
    double sum = 1, u = 1;
    int n = 1;
    const double halfx = x / 2.0;
    do {
      double temp = halfx / (double)n;
      n += 1;
      temp *= temp;
      u *= temp;
      sum += u;
    } while (u >= 1E-21 * sum);
    return sum;
  };
  const double pi = 3.14159265358979232846, frq = 0.5 * 0.90, beta = 6;
  std::vector<double> c((size_t)kNwing);
  c[0] = 2.0 * frq;
  for (int i = 1; i < kNwing; ++i) {
    const double temp = pi * (double)i / (double)kNpc;
    c[(size_t)i] = std::sin(2.0 * temp * frq) / temp;
  }
  const double ibeta = 1.0 / izero(beta), inm1 = 1.0 / ((double)(kNwing - 1));
  for (int i = 1; i < kNwing; ++i) {
    const double temp = (double)i * inm1;
    double temp1 = 1.0 - temp * temp;
    temp1 = (temp1 < 0 ? 0 : temp1);
    c[(size_t)i] *= izero(beta * std::sqrt(temp1)) * ibeta;
  }
  std::vector<float> imp((size_t)kNwing);
  for (int i = 0; i < kNwing; ++i) imp[(size_t)i] = (float)c[(size_t)i];
  float* d = nullptr;
  hipError_t e = hipMalloc((void**)&d, imp.size() * sizeof(float));
  if (e != hipSuccess) return e;
  e = hipMemcpy(d, imp.data(), imp.size() * sizeof(float), hipMemcpyHostToDevice);
  if (e != hipSuccess) { hipFree(d); return e; }
  plan->dev.rs_filter = d;
  return hipSuccess;
}

// bytes per sample of the decoded PCM formats (0: unknown format)
int raw_bytes_per_sample(int format) {
  switch (format) {
    case AFX_RAW_I16: return 2;
    case AFX_RAW_I24: return 3;
    case AFX_RAW_F32: case AFX_RAW_I32: return 4;
    case AFX_RAW_F64: return 8;
    default: return 0;
  }
}

// which of the two statistics kernels a set of series needs (afx_stats.hip)
void stats_regimes(const std::vector<int64_t>& offset, afx::StatsArgs* sa) {
  sa->small_rows = 0;
  sa->need_long = 0;
  for (size_t i = 0; i + 1 < offset.size(); ++i) {
    const int64_t n = offset[i + 1] - offset[i];
    if (n >= 2 && n <= 128) sa->small_rows = std::max<int32_t>(sa->small_rows, (int32_t)n);
    else sa->need_long = 1;
  }
}
void stats_regimes(const afx_batch* b, afx::StatsArgs* sa) { stats_regimes(b->frame_offset, sa); }

// the rhythm tracker's kernels + the statistics of the two onset series (TSampleAnalyser::CalcStatistics covers them)
int run_rhythm(afx_batch* b, hipStream_t stream) {
  if (!(b->mask & AFX_D_RHYTHM) || b->n_bufs == 0) return AFX_OK;
  const afx_plan* plan = b->plan;
  const float rate = (float)plan->desc.sample_rate;
  afx::RhythmArgs ra{};
  ra.pcm = b->d_pcm; ra.pcm_dtype = b->pcm_dtype; ra.n_files = b->n_bufs; ra.files = b->d_rt_files;
  ra.total_frames = b->rt_offset.back(); ra.sample_rate = plan->desc.sample_rate; ra.rayleigh_n = afx::kRayleighTable;
  ra.window = plan->dev.rt_window; ra.tw256 = plan->dev.rt_tw; ra.ut512 = plan->dev.rt_ut; ra.canny = plan->dev.rt_canny;
  ra.rayleigh = plan->dev.rt_rayleigh;
  // TOnsetFftProcessor::SetRelaxTime(25.0f) (OnsetDetector.cpp:105-112; MOnsetWhiteningRelaxTime, RhythmTracker.cpp:21)
  ra.relax_coef = (float)(std::exp((-2.30258509 * (float)128) / ((float)25.0 * rate)));
  ra.norm_complex = (float)(231.70475 / std::pow((double)512, 1.5));   // kFunctionRComplex, OnsetDetector.cpp:300-302
  ra.norm_power = 2560.f / (float)(257 * 512);                          // kFunctionPower, OnsetDetector.cpp:277-279
  ra.thresh[0] = (float)0.2; ra.thresh[1] = (float)0.8;                 // RhythmTracker.cpp:26, 32
  ra.medspan = std::max(3, (int)((rate * (float)0.2) / (float)128 + 0.5f));      // OnsetDetector.cpp:262-264
  ra.mingap[0] = (int)((rate * (float)0.06) / (float)128 + 0.5f);                 // OnsetDetector.cpp:272; RhythmTracker.cpp:28, 34
  ra.mingap[1] = (int)((rate * (float)0.12) / (float)128 + 0.5f);
  if (ra.medspan > 256) return fail(AFX_ERR_UNSUPPORTED, "sample rate too high for the onset detector's median span");
  for (const afx::RhythmFile& rf : b->rt_files)
    if (rf.frames <= afx::kRhythmLdsFrames) ra.lds_frames = std::max(ra.lds_frames, rf.frames);
  ra.odf = b->d_rt_odf; ra.onsets = b->d_rt_onsets; ra.scratch = b->d_rt_scratch; ra.scalars = b->d_rt_scalars;
  ra.n_long = b->rt_n_long; ra.long_rounds = b->rt_long_rounds;
  if (b->rt_n_long > 0) {
    const size_t nl = (size_t)b->rt_n_long;
    ra.long_frame_off = (const int64_t*)b->d_rt_long;
    ra.long_round_off = (const int32_t*)((const unsigned char*)b->d_rt_long + (nl + 1) * 8);
    ra.long_files = (const int32_t*)((const unsigned char*)b->d_rt_long + (nl + 1) * 12);
    ra.long_polar = b->d_rt_polar;
    ra.long_den = (float*)(b->d_rt_polar + b->rt_long_rows * 256);
  }
  HIP_TRY(afx::launch_rhythm(ra, stream));
  if (b->d_rt_stats) {
    afx::StatsArgs sa{};
    sa.rec = b->d_rt_onsets; sa.frame_offset = b->d_rt_foff; sa.n_bufs = b->n_bufs; sa.stride = 2;
    sa.stats = b->d_rt_stats;
    stats_regimes(b->rt_offset, &sa);
    HIP_TRY(afx::launch_stats(sa, stream));
  }
  return AFX_OK;
}

// the rhythm chain on the workspace's side stream, forked here and joined by rhythm_join at the end of the run
// (already_forked: the side stream waits for the batch's stream already -- the time-domain kernels went there first)
int rhythm_fork(afx_batch* b, bool already_forked) {
  if (!(b->mask & AFX_D_RHYTHM) || b->n_bufs == 0) return AFX_OK;
  Workspace& w = *b->ws;
  if (!b->plan->side_stream) return run_rhythm(b, b->stream);
  if (!already_forked) {
    HIP_TRY(hipEventRecord(w.ev_fork, b->stream));
    HIP_TRY(hipStreamWaitEvent(w.side_stream, w.ev_fork, 0));
  }
  const int st = run_rhythm(b, w.side_stream);
  if (st != AFX_OK) return st;
  HIP_TRY(hipEventRecord(w.ev_join, w.side_stream));
  return AFX_OK;
}
int rhythm_join(afx_batch* b) {
  if (!(b->mask & AFX_D_RHYTHM) || b->n_bufs == 0 || !b->plan->side_stream) return AFX_OK;
  HIP_TRY(hipStreamWaitEvent(b->stream, b->ws->ev_join, 0));
  return AFX_OK;
}

bool mask_ok(uint32_t mask) {
  return (mask & ~(uint32_t)AFX_D_STATISTICS) != 0 &&
         !(mask & ~(uint32_t)(AFX_D_ALL_PER_FRAME | AFX_D_MAGNITUDE | AFX_D_STATISTICS | AFX_D_EFFECTIVE_LENGTH | AFX_D_RHYTHM));
}

// first valid buffer's PCM type (the arena of a call is homogeneous); -1 when there is none
int first_valid_dtype(const afx_buf* bufs, int32_t n_bufs) {
  for (int i = 0; i < n_bufs; ++i) {
    const afx_buf& s = bufs[i];
    if (s.n_samples >= 0 && (s.n_samples == 0 || s.pcm) && (s.dtype == AFX_PCM_F32 || s.dtype == AFX_PCM_F64)) return s.dtype;
  }
  return -1;
}

// afx_batch_create with the PCM type of the whole call given (afx_extract_batch cuts large calls into groups: the
// type is decided once, over all buffers, not per group)
int batch_create_typed(afx_plan* plan, const afx_buf* bufs, int32_t n_bufs, uint32_t mask, int dtype, afx_batch** out_batch) {
  if (!plan || !out_batch || n_bufs < 0 || (n_bufs > 0 && !bufs))
    return fail(AFX_ERR_INVALID_ARG, "null argument");
  *out_batch = nullptr;
  if (!mask_ok(mask)) return fail(AFX_ERR_INVALID_ARG, "bad descriptor mask");
  HIP_TRY(hipSetDevice(plan->desc.device));

  // one PCM dtype per batch (the arena is homogeneous); first valid buffer of the call decides
  std::vector<int32_t> status((size_t)n_bufs, AFX_OK);
  std::vector<int64_t> lengths((size_t)n_bufs, 0);
  for (int i = 0; i < n_bufs; ++i) {
    const afx_buf& s = bufs[i];
    const bool good = s.n_samples >= 0 && (s.n_samples == 0 || s.pcm) &&
                      (s.dtype == AFX_PCM_F32 || s.dtype == AFX_PCM_F64);
    if (!good) { status[i] = AFX_ERR_BAD_BUFFER; continue; }
    if (dtype < 0) dtype = s.dtype;
    if (s.dtype != dtype) { status[i] = AFX_ERR_BAD_BUFFER; continue; }
    lengths[i] = s.n_samples;
  }
  if (dtype < 0) dtype = AFX_PCM_F32;
  const size_t esz = (dtype == AFX_PCM_F64) ? 8 : 4;
  auto fill = [&](afx_batch* b) -> int {
    for (int i = 0; i < n_bufs; ++i)
      if (b->used[i] > 0) {
        hipError_t e = hipMemcpyAsync((char*)b->d_pcm + (size_t)b->arena_off[i] * esz, bufs[i].pcm,
                                      (size_t)b->used[i] * esz, hipMemcpyHostToDevice, b->stream);
        if (e != hipSuccess) return hip_fail(e, "hipMemcpy(pcm)");
      }
    return AFX_OK;
  };
  return build_batch(plan, n_bufs, mask, dtype, lengths, status, /*zero_arena=*/false, fill, out_batch);
}

}  // namespace

extern "C" {

int afx_batch_create(afx_plan* plan, const afx_buf* bufs, int32_t n_bufs, uint32_t mask,
                     afx_batch** out_batch) {
  if (!plan || !out_batch || n_bufs < 0 || (n_bufs > 0 && !bufs))
    return fail(AFX_ERR_INVALID_ARG, "null argument");
  return batch_create_typed(plan, bufs, n_bufs, mask, first_valid_dtype(bufs, n_bufs), out_batch);
}

// LoadSample front end (SampleAnalyser.cpp:484-718) on the GPU: decoded interleaved PCM in,
// peak-normalised, silence-trimmed, padded mono doubles in the analysis arena out.
int afx_batch_create_from_raw(afx_plan* plan, const afx_raw* raws, int32_t n_bufs, uint32_t mask,
                              afx_batch** out_batch, afx_load_info* info) {
  if (!plan || !out_batch || n_bufs < 0 || (n_bufs > 0 && !raws))
    return fail(AFX_ERR_INVALID_ARG, "null argument");
  *out_batch = nullptr;
  if (!mask_ok(mask)) return fail(AFX_ERR_INVALID_ARG, "bad descriptor mask");
  HIP_TRY(hipSetDevice(plan->desc.device));

  std::vector<int32_t> status((size_t)n_bufs, AFX_OK);
  std::vector<afx::LoadFile> files((size_t)n_bufs);
  std::vector<afx::ResampleFile> conv;    // files at another rate than the plan's: converted on the GPU first
  std::vector<int32_t> conv_buf;          // their buffer indices
  std::vector<int32_t> file_rate((size_t)n_bufs, 0);
  int64_t raw_bytes = 0;
  for (int i = 0; i < n_bufs; ++i) {
    const afx_raw& r = raws[i];
    const int bps = raw_bytes_per_sample(r.format);
    // SampleAnalyser.cpp:472-482: 1..8 channels, non-empty
    const bool good = bps && r.data && r.n_frames > 0 && r.n_frames < 0x7FFFFFFF && r.channels >= 1 && r.channels <= 8 && r.sample_rate >= 0;
    files[i] = afx::LoadFile{0, 0, 0, 0};
    if (!good) { status[i] = AFX_ERR_BAD_BUFFER; continue; }
    files[i] = afx::LoadFile{raw_bytes, r.n_frames, r.channels, r.format};
    raw_bytes += ((int64_t)r.n_frames * r.channels * bps + 15) & ~(int64_t)15;
  }
  // Sample-rate conversion (SampleAnalyser.cpp:563-607): Speed = file rate / analyser rate in double, the converter runs
  // at factor 1 / Speed and fills NewSizeInSamples = max(1, d2iRound(n / Speed)) samples.  The mono mix and the converted
  // samples of such a file live behind the decoded PCM of the batch in the raw arena; the LoadSample kernels then read
  // the converted samples as a mono file of "16-bit floats".
  int64_t conv_bytes = 0, group_slots = 0, conv_blocks = 0, conv_max_in = 0;
  for (int i = 0; i < n_bufs; ++i) {
    const afx_raw& r = raws[i];
    if (status[i] != AFX_OK || r.sample_rate == 0 || r.sample_rate == plan->desc.sample_rate) continue;
    file_rate[(size_t)i] = r.sample_rate;
    const double speed = (double)r.sample_rate / (double)plan->desc.sample_rate;
    if (speed == 1.0) continue;
    const double factor = 1.0 / speed, scaled = (double)(int)r.n_frames / speed;
    const double n_out_d = std::floor(scaled + 0.5);               // TMath::d2iRound of a positive value (InlineMath.inl:823-826)
    // files above 16 x the analyser's rate (705.6 kHz) are outside what the kernels' zero margins cover; a file whose
    // conversion would blow it up beyond reason (a header claiming a rate below the analyser's / 64, or more than 2^28
    // converted samples = 1.7 hours) fails alone instead of exhausting the device for its whole batch
    if (n_out_d >= 268435456.0 || factor < 1.0 / 16.0 || factor > 64.0 || conv_blocks > 0x7FFFFFF0) { status[i] = AFX_ERR_UNSUPPORTED; files[i] = afx::LoadFile{0, 0, 0, 0}; continue; }
    afx::ResampleFile c{};
    c.raw_off = files[i].raw_off; c.n_in = r.n_frames; c.channels = r.channels; c.format = r.format; c.factor = factor;
    c.n_out = std::max<int64_t>(1, (int64_t)n_out_d);
    c.mono_off = raw_bytes + conv_bytes;
    conv_bytes += ((c.n_in + 2 * afx::kResampleMargin) * 4 + 15) & ~(int64_t)15;
    c.out_off = raw_bytes + conv_bytes;
    conv_bytes += (c.n_out * 4 + 15) & ~(int64_t)15;
    c.group_off = group_slots;                      // one record per 16 output samples
    group_slots += (c.n_out + 15) / 16;
    c.block_off = conv_blocks;
    conv_blocks += afx::resample_blocks(c.n_out);
    conv_max_in = std::max(conv_max_in, c.n_in);
    conv.push_back(c);
    conv_buf.push_back(i);
  }
  const long long t_begin = now_ns();
  // device staging of the decoded PCM and the scan results lives in the batch's pooled workspace
  hipError_t e = hipSuccess;
  Workspace* ws = ws_acquire(plan, &e);
  if (!ws) return hip_fail(e, "workspace");
  hipStream_t s = ws->stream;
  auto bail = [&](int st) { ws_release(plan, ws); return st; };
  unsigned char* d_raw = nullptr;
  afx::LoadFile* d_files = nullptr;
  afx::LoadScan* d_scan = nullptr;
  std::vector<afx::LoadScan> scan((size_t)n_bufs);
  if (n_bufs > 0) {
    if ((e = ws_reserve(ws->raw, (size_t)(raw_bytes + conv_bytes) + 16)) != hipSuccess) return bail(hip_fail(e, "hipMalloc(raw)"));
    if ((e = ws_reserve(ws->files, files.size() * sizeof(afx::LoadFile))) != hipSuccess) return bail(hip_fail(e, "hipMalloc(files)"));
    if ((e = ws_reserve(ws->scan, scan.size() * sizeof(afx::LoadScan))) != hipSuccess) return bail(hip_fail(e, "hipMalloc(scan)"));
    if ((e = ws_reserve(ws->partial, (size_t)n_bufs * afx::load_scan_blocks_per_file(n_bufs) * 16)) != hipSuccess) return bail(hip_fail(e, "hipMalloc(partial)"));
    d_raw = (unsigned char*)ws->raw.p; d_files = (afx::LoadFile*)ws->files.p; d_scan = (afx::LoadScan*)ws->scan.p;
    // Files that lie back to back in host memory, each starting at the next 16-byte boundary (a pipeline's staging
    // buffer), have the layout of the device arena: one transfer moves them all.  Otherwise one transfer per file
    // (each costs ~10 us of runtime overhead, which dominates for thousands of short files).
    bool contiguous = true;
    const char* base = nullptr;
    for (int i = 0; i < n_bufs && contiguous; ++i)
      if (status[i] == AFX_OK) {
        if (!base) base = (const char*)raws[i].data - files[i].raw_off;
        contiguous = ((const char*)raws[i].data == base + files[i].raw_off);
      }
    if (contiguous && base) {
      // up to the last valid file's real end: raw_bytes rounds every file up to 16 bytes, and the bytes behind the
      // caller's last buffer are not the library's to read (a buffer may end at the end of a mapping)
      int64_t used = 0;
      for (int i = 0; i < n_bufs; ++i)
        if (status[i] == AFX_OK)
          used = files[i].raw_off + (int64_t)raws[i].n_frames * raws[i].channels * raw_bytes_per_sample(raws[i].format);
      if ((e = upload_through_plan(plan, ws, d_raw, base, (size_t)used)) != hipSuccess) return bail(hip_fail(e, "hipMemcpy(raw)"));
    } else {
      for (int i = 0; i < n_bufs; ++i)
        if (status[i] == AFX_OK) {
          const int bps = raw_bytes_per_sample(raws[i].format);
          if ((e = hipMemcpyAsync(d_raw + files[i].raw_off, raws[i].data, (size_t)raws[i].n_frames * raws[i].channels * bps,
                                  hipMemcpyHostToDevice, s)) != hipSuccess) return bail(hip_fail(e, "hipMemcpy(raw)"));
        }
    }
    if (!conv.empty()) {
      if ((e = resample_filter_table(plan)) != hipSuccess) return bail(hip_fail(e, "resample filter"));
      if ((e = ws_reserve(ws->rs_files, conv.size() * sizeof(afx::ResampleFile))) != hipSuccess) return bail(hip_fail(e, "hipMalloc(resample files)"));
      if ((e = ws_reserve(ws->rs_groups, (size_t)group_slots * sizeof(afx::ResampleGroup))) != hipSuccess) return bail(hip_fail(e, "hipMalloc(resample groups)"));
      // (pageable source: the copy has left `conv` when the call returns)
      if ((e = hipMemcpyAsync(ws->rs_files.p, conv.data(), conv.size() * sizeof(afx::ResampleFile), hipMemcpyHostToDevice, s)) != hipSuccess) return bail(hip_fail(e, "hipMemcpy(resample files)"));
      if ((e = afx::launch_resample(d_raw, (const afx::ResampleFile*)ws->rs_files.p, (int)conv.size(), conv_blocks, conv_max_in,
                                    (afx::ResampleGroup*)ws->rs_groups.p, plan->dev.rs_filter, s)) != hipSuccess)
        return bail(hip_fail(e, "resample"));
      for (size_t k = 0; k < conv.size(); ++k)
        files[(size_t)conv_buf[k]] = afx::LoadFile{conv[k].out_off, conv[k].n_out, 1, afx::kRawMonoFloat};
    }
    if ((e = hipMemcpyAsync(d_files, files.data(), files.size() * sizeof(afx::LoadFile), hipMemcpyHostToDevice, s)) != hipSuccess) return bail(hip_fail(e, "hipMemcpy(files)"));
    // -48 dB of full scale (MSilenceThresholdDb, SampleAnalyser.cpp:51, 648-649)
    const double silence_floor = 32768.0 * std::exp(-48.0 * (std::log(10.0) / 20.0));
    if ((e = afx::launch_load_scan(d_raw, d_files, n_bufs, silence_floor, ws->partial.p, d_scan, s)) != hipSuccess) return bail(hip_fail(e, "load_scan"));
    if ((e = hipMemcpyAsync(scan.data(), d_scan, scan.size() * sizeof(afx::LoadScan), hipMemcpyDeviceToHost, s)) != hipSuccess) return bail(hip_fail(e, "hipMemcpy(scan)"));
    if ((e = wait_for_stream(ws, s)) != hipSuccess) return bail(hip_fail(e, "hipStreamSynchronize"));
  }
  const long long t_scanned = now_ns();
  // padding rules of SampleAnalyser.cpp:681-701
  const int fft = plan->desc.fft_size;
  std::vector<int64_t> lengths((size_t)n_bufs, 0);
  std::vector<afx::LoadPlace> place((size_t)n_bufs);
  for (int i = 0; i < n_bufs; ++i) {
    place[i] = afx::LoadPlace{};
    if (status[i] != AFX_OK) { if (info) info[i] = afx_load_info{}; continue; }
    // silent leading samples = index of the first sample above the floor (all of them when there is none); the
    // trailing scan stops above that sample (SampleAnalyser.cpp:651-669)
    const int64_t n = files[i].n_frames;
    const int64_t lead = (scan[i].trail < 0) ? n : scan[i].lead, trail = (scan[i].trail < 0) ? 0 : n - 1 - scan[i].trail;
    const int64_t audible = n - lead - trail;
    const int64_t end_pad = ((audible % fft) < fft / 2) ? fft / 2 : 0;
    const int64_t start_pad = (audible + end_pad < fft) ? fft - audible - end_pad : 0;
    lengths[i] = audible + start_pad + end_pad;
    place[i].lead = lead; place[i].audible = audible; place[i].start_pad = start_pad;
    place[i].scaling = scan[i].amplification / 32768.0;      // FinalScaling, SampleAnalyser.cpp:712
    if (info) {
      info[i].peak_value = (float)std::min(1.0, (double)scan[i].max_amp / 32768.0);
      info[i].rms_value = (float)std::min(1.0, std::sqrt(scan[i].sum_sq / (double)n));
      info[i].data_offset = (int32_t)(-lead + start_pad);
      info[i].silent_leading = (int32_t)lead;
      info[i].silent_trailing = (int32_t)trail;
      info[i].reserved = 0;
      info[i].n_samples = lengths[i];
    }
  }
  auto fill = [&](afx_batch* b) -> int {
    hipError_t e2;
    for (int i = 0; i < n_bufs; ++i) place[i].out_off = b->arena_off[i], place[i].out_n = b->used[i];
    if ((e2 = ws_reserve(ws->place, place.size() * sizeof(afx::LoadPlace))) != hipSuccess) return hip_fail(e2, "hipMalloc(place)");
    afx::LoadPlace* d_place = (afx::LoadPlace*)ws->place.p;
    b->h_place = place;   // the batch's own copy: the upload and the write kernel need not be waited for here
    e2 = hipMemcpyAsync(d_place, b->h_place.data(), b->h_place.size() * sizeof(afx::LoadPlace), hipMemcpyHostToDevice, b->stream);
    if (e2 == hipSuccess) e2 = afx::launch_load_write(d_raw, d_files, d_place, n_bufs, (float*)b->d_pcm, b->stream);
    return e2 == hipSuccess ? AFX_OK : hip_fail(e2, "load_write");
  };
  // the workspace (with the staged PCM in it) moves into the batch; build_batch releases it on failure
  std::vector<int64_t> file_samples((size_t)n_bufs, 0);
  std::vector<int32_t> file_offset((size_t)n_bufs, 0);
  // the arena keeps LoadSample's float signal; a sample of TSampleData::mData is (double)float * FinalScaling
  // (SampleAnalyser.cpp:710-718), formed by the kernels as they load it
  std::vector<double> scales((size_t)n_bufs, 1.0);
  for (int i = 0; i < n_bufs; ++i)
    if (status[i] == AFX_OK) {
      scales[(size_t)i] = place[i].scaling;
      file_samples[(size_t)i] = raws[i].n_frames;                                 // mOriginalNumberOfSamples, SampleAnalyser.cpp:464 (before the conversion)
      file_offset[(size_t)i] = (int32_t)(-place[i].lead + place[i].start_pad);    // mDataOffset, SampleAnalyser.cpp:701
    }
  const long long t_placed = now_ns();
  // the decoded PCM has arrived (the scan was waited for): nothing of the caller's is read after this point
  const int st = build_batch(plan, n_bufs, mask, afx::kPcmScaledF32, lengths, status, /*zero_arena=*/false, fill, out_batch, ws, &file_samples,
                             &file_offset, /*wait_for_uploads=*/false, &file_rate, &scales);
  if (g_create_timing.on) {
    const long long t_end = now_ns();
    g_create_timing.ns[0] += t_scanned - t_begin; g_create_timing.ns[1] += t_placed - t_scanned;
    g_create_timing.ns[2] += t_end - t_placed; g_create_timing.ns[3] += 1;
  }
  return st;
}

int afx_batch_fetch_samples(afx_batch* b, int32_t buf, double* dst, int64_t n) {
  if (!b || !dst || buf < 0 || buf >= b->n_bufs || n < 0) return fail(AFX_ERR_INVALID_ARG, "bad argument");
  if (b->pcm_dtype == afx::kPcmF32) return fail(AFX_ERR_INVALID_ARG, "batch does not hold double PCM");
  const int64_t m = std::min<int64_t>(n, b->used[buf]);
  HIP_TRY(hipSetDevice(b->plan->desc.device));
  HIP_TRY(hipStreamSynchronize(b->stream));   // the LoadSample write kernel is not waited for at creation
  if (m <= 0) return AFX_OK;
  if (b->pcm_dtype == afx::kPcmF64) {
    HIP_TRY(hipMemcpy(dst, (const double*)b->d_pcm + b->arena_off[buf], (size_t)m * sizeof(double), hipMemcpyDeviceToHost));
    return AFX_OK;
  }
  // LoadSample batches keep the float signal and the buffer's FinalScaling: the doubles of TSampleData::mData are their
  // products (SampleAnalyser.cpp:710-718), formed here on demand exactly as the kernels form them
  std::vector<float> tmp((size_t)m);
  HIP_TRY(hipMemcpy(tmp.data(), (const float*)b->d_pcm + b->arena_off[buf], (size_t)m * sizeof(float), hipMemcpyDeviceToHost));
  const double scaling = b->buf_scale[(size_t)buf];
  for (int64_t k = 0; k < m; ++k) dst[k] = (double)tmp[(size_t)k] * scaling;
  return AFX_OK;
}

int64_t afx_batch_total_frames(const afx_batch* batch) { return batch ? batch->total_frames : 0; }

int afx_batch_get_info(const afx_batch* b, afx_batch_info* info) {
  if (!b || !info) return fail(AFX_ERR_INVALID_ARG, "null argument");
  const uint32_t fmask = frames_mask(b->mask);
  info->frame_kernel = b->halfwave ? AFX_FRAME_KERNEL_HALFWAVE : AFX_FRAME_KERNEL_WAVE64;
  info->feature_class = b->halfwave ? afx::frames32_class(fmask) : afx::frames_feature_class(fmask);
  info->pcm_kind = b->pcm_dtype;
  info->chunk_frames = b->chunk_frames;
  info->n_chunks = b->n_chunks;
  info->grid_blocks = b->grid_blocks;
  int64_t samples = 0;
  for (size_t i = 0; i < b->used.size(); ++i) samples += (b->used[i] + 3) & ~(int64_t)3;
  info->arena_bytes = samples * (b->pcm_dtype == afx::kPcmF64 ? 8 : 4);
  return AFX_OK;
}

static int batch_run_enqueue(afx_batch* b);

int afx_batch_run(afx_batch* b) {
  if (!b) return fail(AFX_ERR_INVALID_ARG, "null batch");
  HIP_TRY(hipSetDevice(b->plan->desc.device));
  Workspace* const ws = b->ws;
  if (ws && ws->queue_dirty && ws->queue.p) {
    // a run that failed part-way left the work-queue counters and the host's record of them apart (a launch that was
    // counted and never ran): everything of that run is drained, then both start from zero again
    HIP_TRY(hipStreamSynchronize(b->stream));
    if (ws->side_stream) HIP_TRY(hipStreamSynchronize(ws->side_stream));
    HIP_TRY(hipMemsetAsync(ws->queue.p, 0, afx::kQueueSlots * sizeof(unsigned), b->stream));
    for (unsigned& c : ws->queue_count) c = 0;
    ws->queue_dirty = false;
  }
  const int st = batch_run_enqueue(b);
  if (st != AFX_OK && ws) ws->queue_dirty = true;
  return st;
}

static int batch_run_enqueue(afx_batch* b) {
  b->ran = true;
  // The time-domain kernels (autocorrelation, f0, the hop's descriptors) read the PCM only and the spectral chain
  // (STFT, bands) does not need them: they go to the side stream, ahead of the rhythm tracker's, and are joined where
  // the whitening kernel reads what the pitch kernel left (f0, its confidence, the hop's silence flag).  One kernel at a
  // time leaves the tail of every launch to a partly idle chip (a crawl's batch is 10-40 waves per SIMD per kernel).
  const DeviceTables& t = b->plan->dev;
  // the half-wave full classes (2, 3) leave the amplitude of the hop to hop_kernel / pitch_kernel; the magnitude class (4)
  // has the hop in registers at the top of a frame and writes it itself
  const int frame_class = b->halfwave ? afx::frames32_class(frames_mask(b->mask)) : -1;
  const uint32_t post_amplitude = (b->total_frames > 0 && (frame_class == 2 || frame_class == 3))
                                      ? (b->mask & (AFX_D_AMPLITUDE_PEAK | AFX_D_AMPLITUDE_RMS)) : 0u;
  const bool time_work = b->total_frames > 0 && ((b->mask & kTimeBits) || post_amplitude);
  // a kernel's work queue (afx_internal.h): its counter in the workspace and the counter's value when the launch starts
  auto queue_for = [&](int slot, int items) {
    afx::WorkQueue q{b->d_queue ? b->d_queue + slot : nullptr, b->ws->queue_count[slot]};
    if (q.counter) b->ws->queue_count[slot] += (unsigned)items;
    return q;
  };
  // (not with the rhythm tracker selected: its chain, the longer one, has the side stream then -- 21.7 against 21.2 M frames/s
  // on the C4 share with everything selected; without it 34.4 against 33.9 M on C3)
  const bool time_side = time_work && b->plan->side_stream && frames_mask(b->mask) != 0 && !(b->mask & AFX_D_RHYTHM);
  if (time_work) {
    hipStream_t ts = b->stream;
    if (time_side) {
      HIP_TRY(hipEventRecord(b->ws->ev_fork, b->stream));
      HIP_TRY(hipStreamWaitEvent(b->ws->side_stream, b->ws->ev_fork, 0));
      ts = b->ws->side_stream;
    }
    afx::TimeArgs ta{};
    ta.pcm = b->d_pcm; ta.chunks = b->d_chunks; ta.remaining = b->d_rem; ta.n_chunks = b->n_chunks;
    ta.pcm_dtype = b->pcm_dtype; ta.rec = b->d_rec; ta.lay = b->lay;
    ta.t1 = t.t1_f64; ta.t2 = t.t2_f64; ta.post = t.post_f64;
    ta.amplitude = post_amplitude;
    // with f0 selected the pitch kernel has the hop's samples in registers anyway and writes its descriptors too
    const bool want_hop = (b->mask & (AFX_D_AMPLITUDE_SILENCE | AFX_D_AMPLITUDE_ENVELOPE)) || post_amplitude;
    ta.hop_here = (want_hop && (b->mask & AFX_D_F0)) ? 1u : 0u;
    if (want_hop && !ta.hop_here) HIP_TRY(afx::launch_hop(ta, ts));
    if (b->mask & AFX_D_F0) {
      ta.queue = queue_for(afx::kQueuePitch, b->n_chunks);
      HIP_TRY(afx::launch_pitch(ta, ts));
    }
    if (b->mask & AFX_D_AUTO_CORRELATION) {
      ta.queue = queue_for(afx::kQueueAcorr, b->n_chunks);
      HIP_TRY(afx::launch_acorr(ta, ts));
    }
    if (time_side) HIP_TRY(hipEventRecord(b->ws->ev_time_join, ts));
  }
  {
    const int st = rhythm_fork(b, time_side);
    if (st != AFX_OK) return st;
  }
  if (b->d_efflen) {
    // DbToLin(-48 / -24 / -12), AudioMath.inl:108-123
    const double k = std::log(10.0) / 20.0;
    HIP_TRY(afx::launch_effective_length(b->d_pcm, b->pcm_dtype, b->d_spans, b->n_bufs, std::exp(-48.0 * k), std::exp(-24.0 * k),
                                         std::exp(-12.0 * k), b->d_efflen, b->stream));
  }
  if (b->total_frames == 0) {
    // nothing to analyse; empty series still reduce to TStatistics::Calc's Length == 0 result
    if (b->d_stats) {
      afx::StatsArgs sa{};
      sa.rec = b->d_rec; sa.frame_offset = b->d_frame_offset; sa.n_bufs = b->n_bufs; sa.stride = b->lay.stride;
      sa.stats = b->d_stats;
      stats_regimes(b, &sa);
      HIP_TRY(afx::launch_stats(sa, b->stream));
    }
    return rhythm_join(b);
  }
  if (frames_mask(b->mask)) {
    afx::FrameArgs a{};
    a.pcm = b->d_pcm;
    a.chunks = b->d_chunks;
    a.n_chunks = b->n_chunks;
    a.mask = frames_mask(b->mask);
    a.rec = b->d_rec;
    a.lay = b->lay;
    a.mag_out = b->d_mag;
    a.win = t.win; a.t1 = t.t1; a.t2 = t.t2; a.post = t.post; a.melw = t.melw; a.dct = t.dct;
    a.win32 = t.win32; a.tw32 = t.tw32; a.post32 = t.post32; a.melw32 = t.melw32;
#if defined(AFX_STAMPS) && AFX_STAMPS
    // diagnostic build: per-stage cycle counters of the half-wave kernel, printed when the batch is destroyed
    static unsigned long long* stamp_buf = nullptr;
    if (!stamp_buf) { HIP_TRY(hipMalloc((void**)&stamp_buf, (16 + 8192) * 8)); HIP_TRY(hipMemset(stamp_buf, 0, (16 + 8192) * 8)); }
    a.stamps = stamp_buf;
    g_stamp_buf = stamp_buf;
#endif
    if (b->halfwave) {
      a.queue = b->d_queue;
      a.queue_base = b->ws->queue_count[afx::kQueueFrames32];
      b->ws->queue_count[afx::kQueueFrames32] += (unsigned)((b->n_chunks + 1) / 2);
      a.stat_tmp = b->d_stat_tmp;
      a.mag_spare_row = b->total_frames;
      HIP_TRY(afx::launch_frames32(a, b->grid_blocks, b->stream, b->total_frames, b->pcm_dtype == afx::kPcmScaledF32));
    } else HIP_TRY(afx::launch_frames(a, b->plan->desc.precision, b->pcm_dtype, b->grid_blocks, b->stream));
  }
  // the half-wave full class leaves the 28 spectrum bands to bands_kernel (it stores the magnitudes for it)
  const bool bands28_later = b->halfwave && (b->mask & AFX_D_SPECTRUM_BANDS);
  // ... and its magnitude class the spectral statistics
  const bool stats_later = b->halfwave && afx::frames32_class(frames_mask(b->mask)) == 4;
  if ((b->mask & (AFX_D_BAND_FEATURES | AFX_D_SPECTRAL_FLUX)) || bands28_later) {
    afx::BandArgs ba{};
    ba.mag = b->d_mag; ba.chunks = b->d_chunks; ba.n_chunks = b->n_chunks; ba.rec = b->d_rec; ba.lay = b->lay;
    ba.flags = ((b->mask & AFX_D_BAND_FEATURES) ? afx::kBandsFeatures : 0) | ((b->mask & AFX_D_SPECTRAL_FLUX) ? afx::kBandsFlux : 0) |
               (bands28_later ? afx::kBandsSpectrum : 0) | (stats_later ? afx::kBandsStats : 0);
    ba.stat_tmp = b->d_stat_tmp;
    ba.queue = queue_for(afx::kQueueBands, b->n_chunks);
    HIP_TRY(afx::launch_bands(ba, b->stream));
    if (stats_later) {
      afx::FrameArgs fa{};
      fa.mask = frames_mask(b->mask); fa.rec = b->d_rec; fa.lay = b->lay; fa.stat_tmp = b->d_stat_tmp;
      HIP_TRY(afx::launch_stats32_finish(fa, b->stream, b->total_frames));
    }
  }
  if (time_side) HIP_TRY(hipStreamWaitEvent(b->stream, b->ws->ev_time_join, 0));
  if (b->mask & kWhitenBits) {
    afx::WhitenArgs wa{};
    wa.mag = b->d_mag; wa.frame_offset = b->d_frame_offset; wa.n_bufs = b->n_bufs; wa.mask = b->mask;
    wa.chunk_first = b->d_chunk_first; wa.chunks = b->d_wchunks; wa.n_chunks = b->n_wchunks;
    wa.chunk_frames = b->chunk_frames; wa.follower = b->d_follower; wa.need_follow = b->need_follow ? 1 : 0;
    wa.rec = b->d_rec; wa.lay = b->lay;
    // new_aubio_spectral_whitening + set_relax_time(MSpectralWhiteningDecay = 22): awhitening.c:53-87,
    // SampleAnalyser.cpp:44, 805-809
    wa.decay = std::pow(0.001, (double)((float)b->plan->desc.hop_size / (float)b->plan->desc.sample_rate) / 22.0);
    wa.floor_value = 1.e-4;
    wa.queue = queue_for(afx::kQueueWhiten, b->n_wchunks);
    HIP_TRY(afx::launch_whiten(wa, b->stream));
  }
  if (b->d_stats) {
    afx::StatsArgs sa{};
    sa.rec = b->d_rec; sa.frame_offset = b->d_frame_offset; sa.n_bufs = b->n_bufs; sa.stride = b->lay.stride;
    sa.stats = b->d_stats;
    stats_regimes(b, &sa);
    HIP_TRY(afx::launch_stats(sa, b->stream));
  }
  return rhythm_join(b);
}

int afx_batch_sync(afx_batch* b) {
  if (!b) return fail(AFX_ERR_INVALID_ARG, "null batch");
  HIP_TRY(hipStreamSynchronize(b->stream));
  return AFX_OK;
}

int afx_batch_run_timed(afx_batch* b, int32_t steps, float* elapsed_ms) {
  if (!b || steps < 1 || !elapsed_ms) return fail(AFX_ERR_INVALID_ARG, "bad argument");
  HIP_TRY(hipSetDevice(b->plan->desc.device));
  HIP_TRY(hipEventRecord(b->ev0, b->stream));
  for (int i = 0; i < steps; ++i) {
    const int st = afx_batch_run(b);
    if (st != AFX_OK) return st;
  }
  HIP_TRY(hipEventRecord(b->ev1, b->stream));
  HIP_TRY(hipEventSynchronize(b->ev1));
  HIP_TRY(hipEventElapsedTime(elapsed_ms, b->ev0, b->ev1));
  return AFX_OK;
}

int afx_batch_fetch(afx_batch* b, afx_out* out) {
  if (!b || !out) return fail(AFX_ERR_INVALID_ARG, "null argument");
  if (!b->ran) return fail(AFX_ERR_INVALID_ARG, "afx_batch_fetch before afx_batch_run (the pooled workspace would hand back another batch's results)");
  {
    // every requested output must be in the batch mask, whatever the number of frames
    const afx::RecordLayout& lay = b->lay;
    for (const FieldDesc& d : kFields)
      if (out->*(d.out) && lay.*(d.off) < 0) return fail(AFX_ERR_INVALID_ARG, "output requested that is not in the batch mask");
    if (out->magnitude && !(b->mask & AFX_D_MAGNITUDE)) return fail(AFX_ERR_INVALID_ARG, "magnitude not in the batch mask");
    if (out->effective_length && !b->d_efflen && b->n_bufs > 0) return fail(AFX_ERR_INVALID_ARG, "effective_length not in the batch mask");
  }
  HIP_TRY(hipSetDevice(b->plan->desc.device));
  HIP_TRY(hipStreamSynchronize(b->stream));
  if (out->frame_offset) std::memcpy(out->frame_offset, b->frame_offset.data(), b->frame_offset.size() * sizeof(int64_t));
  if (out->buf_status) std::memcpy(out->buf_status, b->buf_status.data(), b->buf_status.size() * sizeof(int32_t));
  if (out->effective_length) {
    if (!b->d_efflen) return fail(AFX_ERR_INVALID_ARG, "effective_length not in the batch mask");
    std::vector<int32_t> lt((size_t)b->n_bufs * 6);
    if (!lt.empty()) HIP_TRY(hipMemcpy(lt.data(), b->d_efflen, lt.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (int32_t i = 0; i < b->n_bufs; ++i)
      for (int j = 0; j < 3; ++j) {
        // TAudioMath::SamplesToMs is float arithmetic (AudioMath.inl:134-137); seconds = ms / 1000.0
        // silent leading samples = index of the first sample above the floor (all of them when there is none);
        // the trailing scan stops above that sample (SampleAnalyser.cpp:1731-1746)
        const int64_t first = lt[(size_t)i * 6 + 2 * j], last = lt[(size_t)i * 6 + 2 * j + 1];
        const int64_t lead = (last < 0) ? b->used[i] : first, trail = (last < 0) ? 0 : b->used[i] - 1 - last;
        const int samples = (int)(b->used[i] - lead - trail);
        const float ms = (float)samples / ((float)b->plan->desc.sample_rate / 1000.0f);
        out->effective_length[(size_t)i * 3 + j] = (b->buf_status[i] == AFX_OK && b->used[i] > 0) ? (double)ms / 1000.0 : 0.0;
      }
  }
  const int64_t F = b->total_frames;
  if (F == 0) return AFX_OK;
  const afx::RecordLayout& l = b->lay;
  struct Field { double* dst; int32_t off; int width; };
  std::vector<Field> fields;
  for (const FieldDesc& d : kFields) fields.push_back(Field{out->*(d.out), l.*(d.off), d.width});
  for (const Field& f : fields)
    if (f.dst && f.off < 0) return fail(AFX_ERR_INVALID_ARG, "output requested that is not in the batch mask");
  if (out->magnitude && !(b->mask & AFX_D_MAGNITUDE)) return fail(AFX_ERR_INVALID_ARG, "magnitude not in the batch mask");
  bool any_series = false;
  for (const Field& f : fields) any_series = any_series || f.dst != nullptr;
  if (l.stride > 0 && any_series) {
    std::vector<double> rec((size_t)F * l.stride);
    HIP_TRY(hipMemcpy(rec.data(), b->d_rec, rec.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (const Field& f : fields) {
      if (!f.dst) continue;
      for (int64_t i = 0; i < F; ++i)
        std::memcpy(f.dst + i * f.width, rec.data() + i * l.stride + f.off, (size_t)f.width * sizeof(double));
    }
  }
  if (out->magnitude)
    HIP_TRY(hipMemcpy(out->magnitude, b->d_mag, (size_t)F * afx::kHalf * sizeof(double), hipMemcpyDeviceToHost));
  return AFX_OK;
}

int afx_batch_fetch_statistics(afx_batch* b, afx_stats_out* out) {
  if (!b || !out) return fail(AFX_ERR_INVALID_ARG, "null argument");
  if (b->n_bufs == 0) return AFX_OK;
  if (!b->d_stats) return fail(AFX_ERR_INVALID_ARG, "AFX_D_STATISTICS was not in the batch mask");
  if (!b->ran) return fail(AFX_ERR_INVALID_ARG, "afx_batch_fetch_statistics before afx_batch_run");
  HIP_TRY(hipSetDevice(b->plan->desc.device));
  HIP_TRY(hipStreamSynchronize(b->stream));
  const afx::RecordLayout& l = b->lay;
  std::vector<double> st((size_t)b->n_bufs * l.stride * 13);
  HIP_TRY(hipMemcpy(st.data(), b->d_stats, st.size() * sizeof(double), hipMemcpyDeviceToHost));
  struct Field { double* dst; int32_t off; int width; };
  std::vector<Field> fields;
  for (const FieldDesc& d : kFields) fields.push_back(Field{out->*(d.stat), l.*(d.off), d.width});
  for (const Field& f : fields) {
    if (!f.dst) continue;
    if (f.off < 0) return fail(AFX_ERR_INVALID_ARG, "statistics requested for a series that is not in the batch mask");
    for (int32_t i = 0; i < b->n_bufs; ++i)
      std::memcpy(f.dst + (size_t)i * f.width * 13, st.data() + ((size_t)i * l.stride + f.off) * 13,
                  (size_t)f.width * 13 * sizeof(double));
  }
  if (out->stats_status)
    for (int32_t i = 0; i < b->n_bufs; ++i) {
      out->stats_status[i] = b->buf_status[i];
    }
  return AFX_OK;
}

int afx_batch_record_layout(const afx_batch* b, int32_t* stride, int32_t* offsets, int32_t* widths) {
  if (!b) return fail(AFX_ERR_INVALID_ARG, "null batch");
  static_assert(sizeof(kFields) / sizeof(kFields[0]) == AFX_NUM_SERIES, "AFX_NUM_SERIES out of date");
  if (stride) *stride = b->lay.stride;
  int i = 0;
  for (const FieldDesc& d : kFields) {
    if (offsets) offsets[i] = b->lay.*(d.off);
    if (widths) widths[i] = d.width;
    ++i;
  }
  return AFX_OK;
}

int afx_batch_fetch_records(afx_batch* b, double* records, double* statistics, int64_t* frame_offset, int32_t* buf_status,
                            double* effective_length) {
  if (!b) return fail(AFX_ERR_INVALID_ARG, "null batch");
  if (!b->ran) return fail(AFX_ERR_INVALID_ARG, "afx_batch_fetch_records before afx_batch_run");
  if (statistics && !b->d_stats && b->n_bufs > 0) return fail(AFX_ERR_INVALID_ARG, "AFX_D_STATISTICS was not in the batch mask");
  HIP_TRY(hipSetDevice(b->plan->desc.device));
  const size_t rec_bytes = (size_t)b->total_frames * b->lay.stride * sizeof(double);
  const size_t stat_bytes = (size_t)b->n_bufs * b->lay.stride * 13 * sizeof(double);
  {
    const Download items[2] = {{records, b->d_rec, rec_bytes}, {statistics, b->d_stats, stat_bytes}};
    HIP_TRY(download_through_plan(b, items, 2));
  }
  if (frame_offset) std::memcpy(frame_offset, b->frame_offset.data(), b->frame_offset.size() * sizeof(int64_t));
  if (buf_status) std::memcpy(buf_status, b->buf_status.data(), b->buf_status.size() * sizeof(int32_t));
  if (effective_length) {
    afx_out tmp = {};
    tmp.effective_length = effective_length;
    const int st = afx_batch_fetch(b, &tmp);
    if (st != AFX_OK) return st;
  }
  return AFX_OK;
}

int afx_batch_set_file_info(afx_batch* b, const afx_file_info* info) {
  if (!b || !info) return fail(AFX_ERR_INVALID_ARG, "null argument");
  if (!(b->mask & AFX_D_RHYTHM)) return fail(AFX_ERR_INVALID_ARG, "AFX_D_RHYTHM was not in the batch mask");
  if (b->n_bufs > 0) {
    set_rhythm_context(b, info);
    HIP_TRY(hipSetDevice(b->plan->desc.device));
    HIP_TRY(hipMemcpyAsync(b->d_rt_files, b->rt_files.data(), b->rt_files.size() * sizeof(afx::RhythmFile), hipMemcpyHostToDevice, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));   // rt_files is pageable host memory that may change again
    b->rt_files_dirty = false;
  }
  return AFX_OK;
}

int64_t afx_batch_rhythm_frames(const afx_batch* b, int64_t* offsets) {
  if (!b || b->rt_offset.empty()) {
    if (b && offsets) std::memset(offsets, 0, ((size_t)b->n_bufs + 1) * sizeof(int64_t));
    return 0;
  }
  if (offsets) std::memcpy(offsets, b->rt_offset.data(), b->rt_offset.size() * sizeof(int64_t));
  return b->rt_offset.back();
}

int afx_batch_fetch_rhythm(afx_batch* b, double* onsets, double* scalars, double* onset_statistics) {
  if (!b) return fail(AFX_ERR_INVALID_ARG, "null batch");
  if (!(b->mask & AFX_D_RHYTHM)) return fail(AFX_ERR_INVALID_ARG, "AFX_D_RHYTHM was not in the batch mask");
  if (!b->ran) return fail(AFX_ERR_INVALID_ARG, "afx_batch_fetch_rhythm before afx_batch_run");
  if (b->n_bufs == 0) return AFX_OK;
  if (onset_statistics && !b->d_rt_stats) return fail(AFX_ERR_INVALID_ARG, "AFX_D_STATISTICS was not in the batch mask");
  HIP_TRY(hipSetDevice(b->plan->desc.device));
  const size_t rows = (size_t)b->rt_offset.back();
  {
    const Download items[3] = {{onsets, b->d_rt_onsets, rows * 2 * sizeof(double)},
                               {scalars, b->d_rt_scalars, (size_t)b->n_bufs * AFX_NUM_RHYTHM_SCALARS * sizeof(double)},
                               {onset_statistics, b->d_rt_stats, (size_t)b->n_bufs * 2 * 13 * sizeof(double)}};
    HIP_TRY(download_through_plan(b, items, 3));
  }
  return AFX_OK;
}

int afx_batch_fetch_onset_functions(afx_batch* b, float* odf) {
  if (!b || !odf) return fail(AFX_ERR_INVALID_ARG, "null argument");
  if (!(b->mask & AFX_D_RHYTHM)) return fail(AFX_ERR_INVALID_ARG, "AFX_D_RHYTHM was not in the batch mask");
  if (!b->ran) return fail(AFX_ERR_INVALID_ARG, "afx_batch_fetch_onset_functions before afx_batch_run");
  HIP_TRY(hipSetDevice(b->plan->desc.device));
  const size_t rows = b->rt_offset.empty() ? 0 : (size_t)b->rt_offset.back();
  if (rows) HIP_TRY(hipMemcpyAsync(odf, b->d_rt_odf, rows * 2 * sizeof(float), hipMemcpyDeviceToHost, b->stream));
  HIP_TRY(hipStreamSynchronize(b->stream));
  return AFX_OK;
}

void afx_batch_destroy(afx_batch* b) {
  if (!b) return;
  afx_plan* const plan = b->plan;
  struct Release { afx_plan* p; ~Release() { plan_release(p); } } release_plan_last{plan};
  hipSetDevice(b->plan->desc.device);
  if (b->stream) hipStreamSynchronize(b->stream);
  // a run that failed between the rhythm chain's fork and its join leaves kernels on the side stream: they must be
  // done before the workspace goes back to the pool
  if (b->ws && b->ws->side_stream) hipStreamSynchronize(b->ws->side_stream);
#if defined(AFX_STAMPS) && AFX_STAMPS
  if (g_stamp_buf) {
    unsigned long long h[16];
    if (hipMemcpy(h, g_stamp_buf, sizeof h, hipMemcpyDeviceToHost) == hipSuccess && h[15]) {
      static const char* names[10] = {"wait DMA+window", "hop+convert", "P1", "exchange", "DMA issue+twiddle", "P2",
                                      "untangle", "mel rows", "win issue+reduce", "log+DCT"};
      unsigned long long tot = 0;
      for (int i = 0; i < 10; ++i) tot += h[i];
      std::fprintf(stderr, "[afx stamps] %llu stamped iterations (2 frames each), %.0f cycles per iteration\n", h[15], (double)tot / h[15]);
      for (int i = 0; i < 10; ++i) std::fprintf(stderr, "[afx stamps] %-18s %8.0f cycles  %5.1f %%\n", names[i], (double)h[i] / h[15], 100.0 * h[i] / tot);
      // lifetimes of every wave of the LAST launch, by wave index inside the workgroup
      static unsigned long long life[8192];
      if (hipMemcpy(life, g_stamp_buf + 16, sizeof life, hipMemcpyDeviceToHost) == hipSuccess) {
        unsigned long long r0 = ~0ull, r1 = 0;
        double sum_core = 0, sum_real = 0;
        double wsum[8] = {}, wmin[8], wmax[8] = {}, wend[8] = {};
        int wn[8] = {};
        for (int w = 0; w < 8; ++w) wmin[w] = 1e30;
        for (int i = 0; i < 2048; ++i) {
          if (!life[4 * i + 1]) continue;
          r0 = std::min(r0, life[4 * i + 2]); r1 = std::max(r1, life[4 * i + 3]);
        }
        for (int i = 0; i < 2048; ++i) {
          if (!life[4 * i + 1]) continue;
          const int w = i & 7;
          const double dr = (double)(life[4 * i + 3] - life[4 * i + 2]) / 100.0;
          sum_core += (double)(life[4 * i + 1] - life[4 * i]); sum_real += dr * 100.0;
          wsum[w] += dr; wmin[w] = std::min(wmin[w], dr); wmax[w] = std::max(wmax[w], dr); ++wn[w];
          wend[w] += (double)(life[4 * i + 3] - r0) / 100.0;
        }
        std::fprintf(stderr, "[afx stamps] last launch: span first-entry..last-exit %.1f us, core clock %.3f GHz\n", (double)(r1 - r0) / 100.0, sum_core / sum_real * 0.1);
        for (int w = 0; w < 8; ++w)
          if (wn[w]) std::fprintf(stderr, "[afx stamps]   wave %d: life mean %.1f us (min %.1f, max %.1f), mean exit at %.1f us\n", w, wsum[w] / wn[w], wmin[w], wmax[w], wend[w] / wn[w]);
      }
      hipMemset(g_stamp_buf, 0, (16 + 8192) * 8);
    }
  }
#endif
  ws_release(b->plan, b->ws);
  delete b;
}

int afx_plan_probe_device(afx_plan* plan) {
  if (!plan) return fail(AFX_ERR_INVALID_ARG, "null plan");
  HIP_TRY(hipSetDevice(plan->desc.device));
  (void)hipGetLastError();
  // a trivial allocation and a query of the plan's own upload stream: both fail with the fault's error once the context is gone
  void* p = nullptr;
  HIP_TRY(hipMalloc(&p, 256));
  HIP_TRY(hipFree(p));
  const hipError_t e = hipStreamQuery(plan->up_stream);
  if (e != hipSuccess && e != hipErrorNotReady) return hip_fail(e, "hipStreamQuery");
  return AFX_OK;
}

int afx_plan_set_blocking_wait(afx_plan* plan, int32_t blocking) {
  if (!plan) return AFX_ERR_INVALID_ARG;
  plan->blocking_wait = blocking != 0;
  return AFX_OK;
}

// Page-locked host memory.  Large blocks are anonymous mappings on 2 MiB boundaries with MADV_HUGEPAGE, touched and then
// registered with the runtime: page-locking huge pages takes a third of hipHostMalloc's time (7.3 vs 21-23 ms per
// 128 MiB on the MI355X box, tools/pin_rate.py) and the first crawl of a process locks ~1.2 GB.  Small blocks, and
// large ones when the mapping or the registration fails, come from hipHostMalloc.
namespace {
std::mutex g_host_mutex;
std::vector<std::pair<void*, size_t>> g_host_mapped;   // registered mappings: base, mapped bytes
constexpr size_t kHugePage = (size_t)2 << 20;
}  // namespace

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

int afx_extract_batch(afx_plan* plan, const afx_buf* bufs, int32_t n_bufs, uint32_t mask, afx_out* out) {
  if (!out) return fail(AFX_ERR_INVALID_ARG, "null output");
  if (!plan || n_bufs < 0 || (n_bufs > 0 && !bufs)) return fail(AFX_ERR_INVALID_ARG, "null argument");
  // Large calls (a crawler handing over 10^5 files) are cut into groups of buffers of at most
  // kSplitFrames frames so that the device workspace (8 KiB of magnitudes per frame when the band
  // descriptors are on) stays bounded; results land at the right rows of the caller's arrays.
  constexpr int64_t kSplitFrames = 1 << 19;
  int64_t total = 0;
  for (int i = 0; i < n_bufs; ++i) total += (bufs[i].n_samples > 0) ? num_frames(plan, bufs[i].n_samples) : 0;
  if (total <= kSplitFrames || n_bufs <= 1) {
    afx_batch* b = nullptr;
    int st = afx_batch_create(plan, bufs, n_bufs, mask, &b);
    if (st != AFX_OK) return st;
    st = afx_batch_run(b);
    if (st == AFX_OK) st = afx_batch_fetch(b, out);
    afx_batch_destroy(b);
    return st;
  }
  const int call_dtype = first_valid_dtype(bufs, n_bufs);
  struct Col { double* afx_out::*field; int width; };
  std::vector<Col> cols;
  for (const FieldDesc& d : kFields) cols.push_back(Col{d.out, d.width});
  cols.push_back(Col{&afx_out::magnitude, 1024});
  int64_t row0 = 0;
  int32_t first = 0;
  if (out->frame_offset) out->frame_offset[0] = 0;
  while (first < n_bufs) {
    int32_t last = first;
    int64_t group = 0;
    while (last < n_bufs) {
      const int64_t f = (bufs[last].n_samples > 0) ? num_frames(plan, bufs[last].n_samples) : 0;
      if (last > first && group + f > kSplitFrames) break;
      group += f;
      ++last;
    }
    afx_out part = *out;
    for (const Col& c : cols)
      if (out->*(c.field)) part.*(c.field) = out->*(c.field) + row0 * c.width;
    std::vector<int64_t> off((size_t)(last - first) + 1);
    part.frame_offset = off.data();
    part.buf_status = out->buf_status ? out->buf_status + first : nullptr;
    part.effective_length = out->effective_length ? out->effective_length + (size_t)first * 3 : nullptr;
    afx_batch* b = nullptr;
    int st = batch_create_typed(plan, bufs + first, last - first, mask, call_dtype, &b);
    if (st != AFX_OK) return st;
    st = afx_batch_run(b);
    if (st == AFX_OK) st = afx_batch_fetch(b, &part);
    afx_batch_destroy(b);
    if (st != AFX_OK) return st;
    if (out->frame_offset)
      for (int32_t i = first; i < last; ++i) out->frame_offset[i + 1] = row0 + off[(size_t)(i - first) + 1];
    row0 += off.back();
    first = last;
  }
  return AFX_OK;
}

}  // extern "C"
