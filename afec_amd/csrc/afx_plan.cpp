// afec_amd/csrc/afx_plan.cpp -- plans: status texts, build info, the constant tables (the analogue of the
// TSampleAnalyser constructor, SampleAnalyser.cpp:162-198: bin range, 2 x Hann window, LibXtract's mel table) and the
// plan's entry points of include/afx.h.  See afx_host.h for the map of the host side.

#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>

#include "afx_host.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

namespace afx {
namespace host {

namespace {
thread_local std::string g_last_error;
}

int fail(int status, const std::string& msg) {
  g_last_error = msg;
  return status;
}
int hip_fail(hipError_t e, const char* what) {
  (void)hipGetLastError();   // the runtime keeps the last error per thread: a later hipGetLastError() check must not see this one
  return fail(e == hipErrorOutOfMemory ? AFX_ERR_OUT_OF_MEMORY : AFX_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
const char* last_error_text() { return g_last_error.c_str(); }

namespace {

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
  HIP_TRY(hipMalloc(&p->dev.probe, 64));   // afx_plan_probe_device's word
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
  hipFree(p->dev.probe);
  p->dev = DeviceTables{};
}

}  // namespace

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

void plan_release(afx_plan* plan) {
  if (plan->refs.fetch_sub(1) != 1) return;
  hipSetDevice(plan->desc.device);
  pool_trim(plan);
  if (plan->up_stream) hipStreamDestroy(plan->up_stream);
  if (plan->down_stream) hipStreamDestroy(plan->down_stream);
  if (plan->probe_stream) hipStreamDestroy(plan->probe_stream);
  free_tables(plan);
  delete plan;
}

}  // namespace host
}  // namespace afx

using namespace afx::host;

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

const char* afx_last_error(void) { return last_error_text(); }

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
  return "afx abi=" "6" " arch=gfx950 stamps=" AFX_INFO_STAMPS " ablation=" AFX_INFO_ABL " src=" AFX_SRC_HASH;
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
      hipStreamCreateWithFlags(&p->down_stream, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&p->probe_stream, hipStreamNonBlocking) != hipSuccess) {
    if (p->up_stream) hipStreamDestroy(p->up_stream);
    if (p->down_stream) hipStreamDestroy(p->down_stream);
    free_tables(p);
    delete p;
    return fail(AFX_ERR_HIP, "hipStreamCreate(copy / probe streams)");
  }
  *out_plan = p;
  return AFX_OK;
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

int afx_device_count(void) {
  int n = 0;
  const hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}

int afx_plan_probe_device(afx_plan* plan) {
  if (!plan) return fail(AFX_ERR_INVALID_ARG, "null plan");
  HIP_TRY(hipSetDevice(plan->desc.device));
  (void)hipGetLastError();
  // Nothing is allocated (after an out-of-memory failure the pooled workspaces still hold their capacity and a
  // hipMalloc may fail on a device that is perfectly alive) and nothing but the probe's own stream is waited for: four
  // bytes written into the plan's probe word, and a query of the upload stream.  Once a fault has taken the context
  // down every one of these calls returns the fault's error.
  HIP_TRY(hipMemsetAsync(plan->dev.probe, 0, 4, plan->probe_stream));
  HIP_TRY(hipStreamSynchronize(plan->probe_stream));
  const hipError_t e = hipStreamQuery(plan->up_stream);
  if (e != hipSuccess && e != hipErrorNotReady) return hip_fail(e, "hipStreamQuery");
  return AFX_OK;
}

int afx_plan_set_blocking_wait(afx_plan* plan, int32_t blocking) {
  if (!plan) return AFX_ERR_INVALID_ARG;
  plan->blocking_wait = blocking != 0;
  return AFX_OK;
}

}  // extern "C"
