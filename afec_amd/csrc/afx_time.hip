// afec_amd/csrc/afx_time.hip -- time-domain neighbours of the spectral loop, gfx950.
//
// Per frame, from the PCM arena (one wave per chunk of consecutive frames, like the frame kernel):
//
//   hop_kernel    amplitude_silence   aubio_silence_detection on the hop, Aubio mathutils.c:345-357, 605-615
//                 amplitude_envelope  TEnvelopeDetector kFast over the hop, SampleAnalyser.cpp (SA) 1787-1804
//   acorr_kernel  auto_correlation    CalcAutoCorrelation, SA:2312-2398 (TAutocorrelation::Calc)
//   pitch_kernel  f0, f0_confidence   aubio yinfast, Aubio pitch/pitchyinfast.c:81-170, pitch.c:399-462,
//                                     SA:876-896 (the fail-safe f0 needs the spectrum: afx_whiten.hip)
//
// The two correlations are computed through the in-register 1024-point complex FFT of afx_fft.h: a real
// 2048-point transform is the complex transform of z[m] = x[2m] + i x[2m+1] plus the even/odd untangle,
// X[k] = E[k] + w^k O[k], X[1024-k] = conj(E[k] - w^k O[k]); the inverse of a Hermitian spectrum runs the
// same steps backwards.  All arithmetic is double whatever the plan's STFT precision.
//
// Layouts: "strided" element = lane + 64 r (FFT in/out), "blocked" element = 16 lane + i (prefix sums,
// first-index searches).  Changes of layout go through the wave's LDS plane with one pad slot per 16.

#include <hip/hip_runtime.h>

#include "afx_internal.h"
#include "afx_device.h"
#include "afx_fft.h"

namespace afx {
namespace {

constexpr int kTimeWaves = 8;
constexpr int kPadSlots = 1344;                 // >= 1024 + 64 pads (blocked layouts), >= kPlaneSlots, >= two staged segments of 640 (acorr_kernel)
constexpr int kTimePlaneBytes = kPadSlots * 8;
// shared tables (t2, post: 16 KiB each; t1: 1 KiB -- from global memory its reads were three exposed cache round trips
// per transform, the vector-memory path being what the frame's own loads wait on), then one plane per wave
constexpr int kLdsT2 = 0, kLdsPost = 16384, kLdsT1 = 32768, kLdsCpow = 32768 + 1024, kLdsEnvScan = 32768 + 1024 + 256,
              kLdsPlanes = 32768 + 1024 + 256 + 1024;
constexpr int kTimeLdsBytes = kLdsPlanes + kTimeWaves * kTimePlaneBytes;

__device__ __forceinline__ int pad_slot(int e) { return e + (e >> 4); }

// x[i] = p[i], i < 16, p 16-byte aligned
__device__ __forceinline__ void load16(const float* p, double (&x)[16]) {
  const float4* q = reinterpret_cast<const float4*>(p);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 v = q[j];
    x[4 * j] = (double)v.x; x[4 * j + 1] = (double)v.y; x[4 * j + 2] = (double)v.z; x[4 * j + 3] = (double)v.w;
  }
}
__device__ __forceinline__ void load16(const double* p, double (&x)[16]) {
  const double2* q = reinterpret_cast<const double2*>(p);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const double2 v = q[j];
    x[2 * j] = v.x; x[2 * j + 1] = v.y;
  }
}

// kPcmScaledF32: the sixteen loaded samples times the buffer's FinalScaling (each product rounded once, like the
// reference's stored doubles: pcm_double, afx_device.h)
template <bool SCALED>
__device__ __forceinline__ void scale16(double (&x)[16], double scale) {
  if (SCALED) {
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = pcm_double<true>(x[i], scale);
  }
}

// sixteen consecutive samples held as they were loaded (the conversion to double happens where they are used: loads
// issued early keep 16 / 32 registers, not 32 / 64)
template <typename TIn>
struct Block16;
template <>
struct Block16<float> {
  float4 q[4];
  __device__ __forceinline__ void load(const float* p) {
    const float4* s = reinterpret_cast<const float4*>(p);
#pragma unroll
    for (int j = 0; j < 4; ++j) q[j] = s[j];
  }
  __device__ __forceinline__ void get(double (&x)[16]) const {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      x[4 * j] = (double)q[j].x; x[4 * j + 1] = (double)q[j].y; x[4 * j + 2] = (double)q[j].z; x[4 * j + 3] = (double)q[j].w;
    }
  }
};
template <>
struct Block16<double> {
  double2 q[8];
  __device__ __forceinline__ void load(const double* p) {
    const double2* s = reinterpret_cast<const double2*>(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) q[j] = s[j];
  }
  __device__ __forceinline__ void get(double (&x)[16]) const {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      x[2 * j] = q[j].x; x[2 * j + 1] = q[j].y;
    }
  }
};

// lane 0's value, in every lane
__device__ __forceinline__ float first_lane(float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)); }
__device__ __forceinline__ double first_lane(double v) { return read_lane<0>(v); }

using mask64 = unsigned long long;

// aubio_silence_detection: 10 log10(mean x^2) < -48 dB (mathutils.c:605-615, MSilenceThresholdDb).  log10 is
// monotonic, so the test is mean x^2 < 10^-4.8 (they can only disagree for a level within an ulp of the
// threshold); a NaN level is "not silent" in both forms.
__device__ __forceinline__ bool au_silent(double sum_sq, int n) {
  return sum_sq / (double)n < 1.5848931924611134e-05;
}

// The descriptors of a hop (the first 1024 samples of a frame) from its samples in the blocked layout, x[i] = sample
// 16 lane + i, and their sum of squares e: the silence flag, the envelope maximum (TEnvelopeDetector kFast, reset per
// frame: SA:1787-1804) and, when `amplitude` asks for them, CalcAmplitudePeak / CalcAmplitudeRms (SA:1760-1783).
// env_i = in_i + c (env_{i-1} - in_i) is affine in env_{i-1} with slope c: a lane runs its 16 samples from 0, the carries
// are an affine scan over the lanes (slope c^16), and env_i = local_i + c^(i+1) carry.
__device__ __forceinline__ double hop_energy(const double (&x)[16]) {
  double e = 0.0;
#pragma unroll
  for (int i = 0; i < 16; ++i) e += x[i] * x[i];
  return wave_sum(e);
}
// lane constants of the envelope's prefix scan (hop_descriptors): coef^(16 (j + 1)) takes the end value of the last lane
// of the row before to lane j of a row; from lane 31 to lane j of row 2 it is the same power, of row 3 sixteen lanes more
struct EnvelopeScan { double from_row_before, from_lane31; };
__device__ __forceinline__ EnvelopeScan envelope_scan(double coef, int lane) {
  const int j = lane & 15;
  EnvelopeScan e;
  e.from_row_before = pow(coef, (double)(16 * (j + 1)));
  e.from_lane31 = (lane >= 48) ? pow(coef, (double)(16 * (j + 17))) : e.from_row_before;
  return e;
}
// cpow[i] = coef^i, i <= 16 (registers in hop_kernel, an LDS table in pitch_kernel)
template <typename Pow>
__device__ __forceinline__ void hop_descriptors(const double (&x)[16], double e, double coef, const Pow& cpow, const EnvelopeScan& scan,
                                                uint32_t amplitude, const RecordLayout& lay, double* rec, int lane) {
  if (amplitude) {
    double peak = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) peak = fmax(peak, fabs(x[i]));
    peak = wave_max(peak);
    if (lane == 0) {
      if ((amplitude & (1u << 11)) && lay.amp_peak >= 0) rec[lay.amp_peak] = peak;
      if ((amplitude & (1u << 12)) && lay.amp_rms >= 0) rec[lay.amp_rms] = nan_to_zero(sqrt(e / (double)kHop));
    }
  }
  if (lay.silence < 0 && lay.envelope < 0) return;
  double env = 0.0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const double in = fabs(x[i]);
    env = in + coef * (env - in);        // TEnvelopeDetector::Run, Envelopes.inl:14-18
  }
  // The envelope at the end of lane l's sixteen samples is C_l = env_l + s C_(l-1), s = coef^16: a prefix "sum" with a
  // factor per step.  Inside the rows of 16 lanes with DPP row shifts (factors s, s^2, s^4, s^8), then the last lane of
  // row 0 into row 1 and of row 2 into row 3, then lane 31 into rows 2 and 3, each times the power of s that separates the
  // two lanes (EnvelopeScan) -- no LDS round trips (six ds_bpermute steps before).
  double carry = env;
  const double s1 = cpow[16], s2 = s1 * s1, s4 = s2 * s2, s8 = s4 * s4;
  carry = fma(s1, dpp_or_zero<kDppRowShr1>(carry), carry);
  carry = fma(s2, dpp_or_zero<kDppRowShr2>(carry), carry);
  carry = fma(s4, dpp_or_zero<kDppRowShr4>(carry), carry);
  carry = fma(s8, dpp_or_zero<kDppRowShr8>(carry), carry);
  carry = fma(scan.from_row_before, dpp_or_zero<kDppRowBcast15, 0xA>(carry), carry);
  carry = fma(scan.from_lane31, dpp_or_zero<kDppRowBcast31, 0xC>(carry), carry);
  const double in_carry = dpp_or_zero<0x138>(carry);      // wave_shr:1: lane l reads lane l - 1, lane 0 gets 0
  // the lane's sixteen values once more, now from the state the lanes before it leave: the reference's recurrence itself
  // (the first pass, from 0, only gave the scan its per-lane term; its sixteen values are not kept -- 32 registers that
  // the pitch kernel, which calls this with its transforms' registers live, does not have)
  double top = 0.0;
  env = in_carry;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const double in = fabs(x[i]);
    env = in + coef * (env - in);
    top = fmax(top, env);
  }
  top = wave_max(top);
  if (lane == 0) {
    if (lay.silence >= 0) rec[lay.silence] = au_silent(e, kHop) ? 1.0 : 0.0;   // SA:865-868
    if (lay.envelope >= 0) rec[lay.envelope] = top;
  }
}

// ---------------------------------------------------------------------------------------------
// hop kernel: silence flag + envelope maximum of the hop (first 1024 samples of the frame)
// ---------------------------------------------------------------------------------------------
template <typename TIn, bool SCALED>
__global__ __launch_bounds__(256) void hop_kernel(const TimeArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int wave_stride = gridDim.x * 4;
  const TIn* const pcm = reinterpret_cast<const TIn*>(a.pcm);

  // TEnvelopeDetector(kFast, MEnvelopeTimeInMs = 8, rate): mCoef = pow(0.01, 1000 / (ms rate)), Envelopes.cpp:55-70.
  // env_i = in_i + c (env_{i-1} - in_i) is affine in env_{i-1} with slope c: a lane runs its 16 samples from
  // 0, the carries are an affine scan over the lanes (slope c^16), and env_i = local_i + c^(i+1) carry.
  const double coef = pow(0.01, 1000.0 / (8.0 * (double)kSampleRate));
  double cpow[17];
  cpow[0] = 1.0;
#pragma unroll
  for (int i = 1; i <= 16; ++i) cpow[i] = cpow[i - 1] * coef;
  const EnvelopeScan env_scan = envelope_scan(coef, lane);

  for (int ci = wave_global; ci < a.n_chunks; ci += wave_stride) {
    const Chunk ch = a.chunks[ci];
    const double sc = SCALED ? wave_uniform(ch.scale) : 1.0;
    for (int fi = 0; fi < ch.nframes; ++fi) {
      double x[16];
      load16(pcm + ch.sample_off + (int64_t)fi * kHop + 16 * lane, x);
      scale16<SCALED>(x, sc);
      double* const rec = a.rec + ((int64_t)ch.frame0 + fi) * a.lay.stride;

      hop_descriptors(x, hop_energy(x), coef, cpow, env_scan, a.amplitude, a.lay, rec, lane);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// FFT plumbing shared by the correlation kernels
// ---------------------------------------------------------------------------------------------
struct TimeCtx {
  const cx<double>* t1;     // LDS, + 16 (lane >> 4)
  const cx<double>* t2;     // LDS, + lane
  const cx<double>* post;   // LDS, + lane: w2048^(lane + 64 r) at [64 r]
  unsigned char* plane;
  double* plane_d;
  unsigned plane_rd_addr;
  int lane, partner;
};

__device__ __forceinline__ TimeCtx time_setup(const TimeArgs& a, unsigned char* lds_raw) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  copy_lds_table(lds_raw + kLdsT2, a.t2, 16384, threadIdx.x, kTimeWaves * 64);
  copy_lds_table(lds_raw + kLdsPost, a.post, 16384, threadIdx.x, kTimeWaves * 64);
  copy_lds_table(lds_raw + kLdsT1, a.t1, 1024, threadIdx.x, kTimeWaves * 64);
  __syncthreads();
  TimeCtx c;
  c.t1 = reinterpret_cast<const cx<double>*>(lds_raw + kLdsT1) + 16 * (lane >> 4);
  c.t2 = reinterpret_cast<const cx<double>*>(lds_raw + kLdsT2) + lane;
  c.post = reinterpret_cast<const cx<double>*>(lds_raw + kLdsPost) + lane;
  c.plane = lds_raw + kLdsPlanes + wave * kTimePlaneBytes;
  c.plane_d = reinterpret_cast<double*>(c.plane);
  c.plane_rd_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds_raw + kLdsPlanes +
                    wave * kTimePlaneBytes + 8u * lane;
  c.lane = lane;
  c.partner = (64 - lane) & 63;
  return c;
}

__device__ __forceinline__ void fft(cx<double> (&v)[16], const TimeCtx& c) {
  fft1024<double>(v, c.t1, c.t2, c.plane, c.plane_rd_addr, c.lane);
}

// Z[1024 - k] for k = lane + 64 r (Z is 1024-periodic)
__device__ __forceinline__ cx<double> partner_of(const cx<double> (&v)[16], int r, const TimeCtx& c) {
  cx<double> p{__shfl(v[15 - r].re, c.partner), __shfl(v[15 - r].im, c.partner)};
  if (c.lane == 0) p = v[(16 - r) & 15];
  return p;
}

// even/odd parts (times 2) of the 2048-point spectrum of the real sequence packed into z:
// 2 E[k] = Z[k] + conj(Z[1024-k]),  2 O[k] = -i (Z[k] - conj(Z[1024-k]))
__device__ __forceinline__ void even_odd(cx<double> z, cx<double> p, cx<double>& e, cx<double>& o) {
  e = {z.re + p.re, z.im - p.im};
  o = {z.im + p.im, p.re - z.re};
}

// ---------------------------------------------------------------------------------------------
// auto_correlation (SA:2312-2398)
// ---------------------------------------------------------------------------------------------
// Two frames per pair of transforms.  TAutocorrelation::Calc (Autocorrelation.cpp:84-104) of a segment of w <= 529
// samples is the inverse transform of its power spectrum.  With circular length N = 1024 the lags k <= N - w come out
// clean (c[k] = r[k] + r[N - k], r[m] = 0 for m >= w); the few lags above that (k in [w - 32, w), at most 32 terms each)
// are summed directly, and the one lag that can still be polluted (k = 496 for w = 529: its alias is r[528]) is
// corrected with the directly summed value.  Two real sequences share one complex transform, z = a + i b:
// 2 A[k] = Z[k] + conj(Z[N-k]), 2 B[k] = -i (Z[k] - conj(Z[N-k])), and since the power spectra are real and even,
// the forward transform of |2A|^2 + i |2B|^2 is 4 N (r_a + i r_b).  (Before: each frame's zero-padded 2048-point real
// transform and its inverse: two complex 1024-point transforms per frame instead of per pair.)
template <typename TIn, bool SCALED>
__global__ __launch_bounds__(kTimeWaves * 64) void acorr_kernel(const TimeArgs a) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  const TimeCtx c = time_setup(a, lds_raw);
  const int lane = c.lane;
  const int wave_global = blockIdx.x * kTimeWaves + (threadIdx.x >> 6);
  const int wave_stride = gridDim.x * kTimeWaves;
  const TIn* const pcm = reinterpret_cast<const TIn*>(a.pcm);
  // TAudioMath::MsToSamples(44100, 0.8f) = 35, (44100, 12.0f) = 529 (float maths, AudioMath.inl:127-130)
  constexpr int kMinPeriod = 35, kSeekWidth = 529, kMaxSeek = kFft / 2, kN = 1024, kDirect = 32;
  constexpr int kSeg = 640;           // staged samples per frame: rows start/64 .. start/64 + 9
  constexpr double kScale = 1.0 / 4096.0;   // the transforms return 4 N r

  for (int ci = wave_global; ci < a.n_chunks; ci = next_item(a.queue, ci, min(wave_stride, a.n_chunks), wave_stride, lane)) {
    const Chunk ch = a.chunks[ci];
    const double sc = SCALED ? wave_uniform(ch.scale) : 1.0;
    const int remaining0 = a.remaining[ci];
    // Everything the two searches and the correlations of a pair of frames may touch -- samples 0 .. 3135 of the first
    // frame, as far as the buffer has them (the arena keeps up to 64 samples past the last frame; zeros behind it) -- is
    // loaded in one go: rw[q] = x[64 q + lane]; the second frame's rows are rw[16 ..].  "x[p+1] > x[p]" is one ballot
    // per row -- the row against itself shifted by one lane (wave_shl:1), lane 63 against lane 0 of the next row -- and
    // the first-index searches are scalar bit scans; no load depends on a search result.  The rows of the next pair are
    // asked for as soon as this pair has staged its segments: they travel while the two transforms run.
    constexpr int kRowsFrame = 33, kRowsPair = 49;
    TIn rw[kRowsPair];
    auto load_rows = [&](int fi) {
      const TIn* const x = pcm + ch.sample_off + (int64_t)fi * kHop;
      const bool second = fi + 1 < ch.nframes;
      // the second frame's row 32 ends its own limit; without a second frame the first frame's row 32 does
      const int lim = min(remaining0 - fi * kHop, second ? kHop + 64 * kRowsFrame : 64 * kRowsFrame);
#pragma unroll
      for (int q = 0; q < kRowsPair; ++q) {
        const int p = 64 * q + lane;
        rw[q] = (q < 32 || p < lim) ? ((q < kRowsFrame || second) ? x[q < kRowsFrame || second ? p : 0] : (TIn)0) : (TIn)0;
      }
    };
    // (double PCM -- buffers a caller normalised itself -- would hold 98 registers of rows in flight: loaded per pair)
    constexpr bool kPrefetch = sizeof(TIn) == 4;
    if (kPrefetch) load_rows(0);
    for (int fi = 0; fi < ch.nframes; fi += 2) {
      if (!kPrefetch) load_rows(fi);
      const bool have_b = fi + 1 < ch.nframes;
      // per frame of the pair (s = 0, 1): the two searches and the segment; wave-uniform results
      int start[2], period[2], width[2], from[2], off[2];
      bool active[2];
      wave_lds_fence();
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int row0 = 16 * s2;                          // the frame's first row in rw
        int remaining = remaining0 - (fi + s2) * kHop;     // mData.Size() - n, >= 2048 for an emitted frame
        auto rising = [&](int q) -> mask64 {               // bit l: x[64 q + l + 1] > x[64 q + l] of this frame
          const TIn next0 = (q + 1 < kRowsFrame) ? first_lane(rw[row0 + (q + 1 < kRowsFrame ? q + 1 : q)]) : (TIn)0;
          return __ballot(next_lane(rw[row0 + q], next0) > rw[row0 + q]);
        };
        int st = 0;
        if (s2 == 0 || have_b) {
          // first rising step in [0, min(remaining, 1024) - 1)  (SA:2328-2341)
          const int bound = min(remaining, kMaxSeek) - 1;
          bool found = false;
#pragma unroll
          for (int q = 0; q < kMaxSeek / 64; ++q) {
            if (!found && 64 * q < bound) {
              mask64 m = rising(q);
              const int valid = bound - 64 * q;
              if (valid < 64) m &= ((mask64)1 << valid) - 1;
              if (m) { st = 64 * q + __ffsll((long long)m) - 1; found = true; }
            }
          }
          if (found) remaining -= st;
        }
        // next rising step after the minimum period (SA:2343-2356): positions [lo, hi) of the frame
        const int seek_off = min(remaining, kMinPeriod);
        int per = seek_off;
        if (s2 == 0 || have_b) {
          const int bound = min(remaining - seek_off, kMaxSeek) - 1;
          const int lo = st + seek_off, hi = lo + bound;
          bool found = false;
#pragma unroll
          for (int q = 0; q < kRowsFrame; ++q) {
            if (!found && 64 * q + 63 >= lo && 64 * q < hi) {
              mask64 m = rising(q);
              if (64 * q < lo) m &= ~(((mask64)1 << (lo - 64 * q)) - 1);
              if (hi - 64 * q < 64) m &= ((mask64)1 << (hi - 64 * q)) - 1;
              if (m) { per = seek_off + (64 * q + __ffsll((long long)m) - 1 - lo); found = true; }
            }
          }
        }
        start[s2] = st; period[s2] = per;
        active[s2] = (s2 == 0 || have_b) && remaining != 0 && per < remaining;
        width[s2] = min(remaining, kSeekWidth);
        from[s2] = per / 2;
        off[s2] = st & 63;
        // the segment x[start .. start + width) sits in rows start/64 .. start/64 + 9: through the plane (zeros for a
        // frame that has no correlation: its half of the transform's input must be clean)
        const int q0 = st >> 6;
        double* const seg = c.plane_d + kSeg * s2;
#pragma unroll
        for (int q = 0; q < 25; ++q)
          if (q >= q0 && q < q0 + 10) seg[64 * (q - q0) + lane] = active[s2] ? pcm_double<SCALED>(rw[row0 + q], sc) : 0.0;
      }
      wave_lds_fence();
      // the rows are dead: the next pair's take their registers (the last pair of a chunk re-reads its own)
      if (kPrefetch) {
        __builtin_amdgcn_sched_barrier(0);
        load_rows(fi + 2 < ch.nframes ? fi + 2 : fi);
        __builtin_amdgcn_sched_barrier(0);
      }
      double best[2] = {0.0, 0.0};
      if (active[0] || active[1]) {
        // ---- the lags the circular transform does not give cleanly, summed directly: lane l' = lane & 31 of half
        //      s = lane >> 5 takes k = w - 32 + l' (32 - l' terms) ----
        const int hs = lane >> 5, lp = lane & 31;
        const int w_h = hs ? width[1] : width[0], off_h = hs ? off[1] : off[0];
        const double* const seg_h = c.plane_d + kSeg * hs + off_h;
        const int k_h = w_h - kDirect + lp;
        double direct = 0.0;
        {
          const int kk = k_h >= 0 ? k_h : 0;
#pragma unroll 8
          for (int j = 0; j < kDirect; ++j) {
            const double u = seg_h[j], v2 = seg_h[j + kk];     // (within the staged 640: off + 31 + 528 < 640)
            direct = (j < kDirect - lp) ? fma(u, v2, direct) : direct;
          }
          if (k_h < 0) direct = 0.0;
        }
        // ---- z = a + i b, m = lane + 64 r (zero from w on) ----
        cx<double> v[16];
        double mxa = 0.0, mxb = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = 64 * r + lane;
          v[r] = {0.0, 0.0};
          if (r < 9) {
            v[r] = {(m < width[0]) ? c.plane_d[off[0] + m] : 0.0, (m < width[1]) ? c.plane_d[kSeg + off[1] + m] : 0.0};
            mxa = fmax(mxa, fabs(v[r].re));
            mxb = fmax(mxb, fabs(v[r].im));
          }
        }
        wave_lds_fence();
        // The two frames share one complex transform, whose rounding errors are relative to the larger of the two: each
        // segment is brought to [0.5, 1) by a power of two first (exact, and the result -- a ratio of two of its
        // correlation values -- does not see it).  A segment of zeros has r[0] = 0 and the result 0
        // (Autocorrelation.cpp:97-105), not rounding noise of its partner over rounding noise.
        mxa = wave_max(mxa);
        mxb = wave_max(mxb);
        const bool live_a = active[0] && mxa > 0.0, live_b = active[1] && mxb > 0.0;
        const double ua = ldexp(1.0, -__builtin_amdgcn_frexp_exp(mxa > 0.0 ? mxa : 1.0));
        const double ub = ldexp(1.0, -__builtin_amdgcn_frexp_exp(mxb > 0.0 ? mxb : 1.0));
#pragma unroll
        for (int r = 0; r < 9; ++r) v[r] = {v[r].re * ua, v[r].im * ub};
        direct *= hs ? ub * ub : ua * ua;
        fft(v, c);
        cx<double> g[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const cx<double> p = partner_of(v, r, c);
          cx<double> e, o;
          even_odd(v[r], p, e, o);                       // 2 A[k], 2 B[k]
          g[r] = {e.re * e.re + e.im * e.im, o.re * o.re + o.im * o.im};
        }
        fft(g, c);
        // g[r] = 4 N (r_a[m] + i r_b[m]), m = lane + 64 r
        const double r0a = read_lane<0>(g[0].re) * kScale, r0b = read_lane<0>(g[0].im) * kScale;
        // the alias of lag N - (w - 1) when that lag lies below the directly summed ones: r[w - 1], lane l' = 31's sum
        const double last_a = read_lane<31>(direct), last_b = read_lane<63>(direct);
        double top_a = -1.0e300, top_b = -1.0e300;
#pragma unroll
        for (int r = 0; r < 9; ++r) {
          const int m = 64 * r + lane;
          const double ca = g[r].re * kScale - ((m == kN - (width[0] - 1)) ? last_a : 0.0);
          const double cb = g[r].im * kScale - ((m == kN - (width[1] - 1)) ? last_b : 0.0);
          if (m >= from[0] && m < width[0] - kDirect) top_a = fmax(top_a, ca);
          if (m >= from[1] && m < width[1] - kDirect) top_b = fmax(top_b, cb);
        }
        // the directly summed lags of each half
        const int from_h = hs ? from[1] : from[0];
        const double dtop = (k_h >= 0 && k_h >= from_h) ? direct : -1.0e300;
        top_a = fmax(top_a, hs ? -1.0e300 : dtop);
        top_b = fmax(top_b, hs ? dtop : -1.0e300);
        top_a = wave_max(top_a);
        top_b = wave_max(top_b);
        if (r0a != 0.0) top_a /= r0a;                    // Autocorrelation.cpp:97-105
        if (r0b != 0.0) top_b /= r0b;
        best[0] = live_a ? fmax(0.0, top_a) : 0.0;
        best[1] = live_b ? fmax(0.0, top_b) : 0.0;
      }
      if (lane == 0) {
        a.rec[((int64_t)ch.frame0 + fi) * a.lay.stride + a.lay.autocorr] = best[0];
        if (have_b) a.rec[((int64_t)ch.frame0 + fi + 1) * a.lay.stride + a.lay.autocorr] = best[1];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// f0 + confidence: aubio yinfast on the 2048-sample frame
// ---------------------------------------------------------------------------------------------
template <typename TIn>
struct PairOf;
template <>
struct PairOf<float> { using type = float2; };
template <>
struct PairOf<double> { using type = double2; };

template <typename TIn, bool SCALED>
__global__ __launch_bounds__(kTimeWaves * 64) void pitch_kernel(const TimeArgs a) {
  using Pair = typename PairOf<TIn>::type;
  extern __shared__ __align__(16) unsigned char lds_raw[];
  const TimeCtx c = time_setup(a, lds_raw);
  const int lane = c.lane;
  const int wave_global = blockIdx.x * kTimeWaves + (threadIdx.x >> 6);
  const int wave_stride = gridDim.x * kTimeWaves;
  const TIn* const pcm = reinterpret_cast<const TIn*>(a.pcm);
  constexpr int W = kFft / 2;
  constexpr double kTol = 0.75;                 // MPitchTolerance, SA:62, 801
  // TEnvelopeDetector(kFast, MEnvelopeTimeInMs = 8, rate): mCoef = pow(0.01, 1000 / (ms rate)), Envelopes.cpp:55-70; its
  // powers in LDS (hop_descriptors)
  const double env_coef = pow(0.01, 1000.0 / (8.0 * (double)kSampleRate));
  double* const env_pow = reinterpret_cast<double*>(lds_raw + kLdsCpow);
  EnvelopeScan* const env_scan_of_lane = reinterpret_cast<EnvelopeScan*>(lds_raw + kLdsEnvScan);
  if (a.hop_here && threadIdx.x == 0) {
    double pw = 1.0;
    for (int i = 0; i <= 16; ++i) { env_pow[i] = pw; pw *= env_coef; }
  }
  if (a.hop_here && threadIdx.x < 64) env_scan_of_lane[threadIdx.x] = envelope_scan(env_coef, threadIdx.x);
  __syncthreads();
  constexpr int kBig = 1 << 30;

  for (int ci = wave_global; ci < a.n_chunks; ci = next_item(a.queue, ci, min(wave_stride, a.n_chunks), wave_stride, lane)) {
    const Chunk ch = a.chunks[ci];
    const double sc = SCALED ? wave_uniform(ch.scale) : 1.0;
    // transform of the zero-padded first half of the chunk's first frame (packed z[m] = x[2m] + i x[2m+1])
    cx<double> zu[16];
    {
      const Pair* src = reinterpret_cast<const Pair*>(pcm + ch.sample_off) + lane;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        zu[r] = {0.0, 0.0};
        if (r < 8) {
          const Pair p = src[64 * r];
          zu[r] = {pcm_double<SCALED>(p.x, sc), pcm_double<SCALED>(p.y, sc)};
        }
      }
    }
    // The loads of a frame are issued long before they are used (two waves per SIMD hide little: every exposed load
    // was ~a microsecond of an idle vector ALU): the second half of frame fi + 1 -- the only new samples of a frame --
    // travels during the whole of frame fi (nx), the blocked copies for the squared-difference terms during the
    // inverse transform.
    // (double PCM -- buffers a caller normalised itself -- holds twice the registers in flight: loaded where they are used)
    constexpr bool kPrefetch = sizeof(TIn) == 4;
    Pair nx[8];
    if (kPrefetch) {
      const Pair* src = reinterpret_cast<const Pair*>(pcm + ch.sample_off + W) + lane;
#pragma unroll
      for (int r = 0; r < 8; ++r) nx[r] = src[64 * r];
    }
    fft(zu, c);
    for (int fi = 0; fi < ch.nframes; ++fi) {
      const TIn* const x = pcm + ch.sample_off + (int64_t)fi * kHop;
      double* const rec = a.rec + ((int64_t)ch.frame0 + fi) * a.lay.stride;

      // ---- r_t(tau) = sum_{j<W} x[j] x[j+tau], tau < W, as conj(U) X: U = spectrum of the zero-padded first
      //      half, X = spectrum of the frame = U + (-1)^k Un with Un the spectrum of the zero-padded second half.
      //      The second half is the next frame's first half, so each frame costs one forward transform. ----
      cx<double> zn[16];
      if (!kPrefetch) {
        const Pair* src = reinterpret_cast<const Pair*>(x + W) + lane;
#pragma unroll
        for (int r = 0; r < 8; ++r) nx[r] = src[64 * r];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        zn[r] = {0.0, 0.0};
        if (r < 8) zn[r] = {pcm_double<SCALED>(nx[r].x, sc), pcm_double<SCALED>(nx[r].y, sc)};
      }
      if (kPrefetch) {
        __builtin_amdgcn_sched_barrier(0);
        // the next frame's second half (the last frame of a chunk re-reads its own: no branch, nothing is read
        // beyond what the chunk's frames cover)
        const int fn = (fi + 1 < ch.nframes) ? fi + 1 : fi;
        const Pair* src = reinterpret_cast<const Pair*>(pcm + ch.sample_off + (int64_t)fn * kHop + W) + lane;
#pragma unroll
        for (int r = 0; r < 8; ++r) nx[r] = src[64 * r];
        __builtin_amdgcn_sched_barrier(0);
      }
      fft(zn, c);
      const double sign = (lane & 1) ? -1.0 : 1.0;      // (-1)^k, k = lane + 64 r (1024 - k has the same parity)
      // C = conj(U) X at k and at 1024-k, then the packed spectrum of the inverse: Zc = Ec + i Oc,
      // Ec = (C + conj(C'))/2, Oc = (C - conj(C'))/2 conj(w^k).  The factors 1/2 are carried along instead of
      // applied (2E, 2O, 2U, 2X, 4C, 8 Zc: exact) and undone with the 1/1024 of the inverse.
      // Bins k and 1024 - k share every intermediate (U[k'] = Up, X[k'] = Xp, the roles of C and Cp swap, w[k'] = -conj(w)):
      // the output at k' is {ec.re + oc.im, ec.im - oc.re} where the output at k is {ec.re - oc.im, -(ec.im + oc.re)}.  A lane
      // therefore works on half of its bins (registers 0..7) and hands the partner lane the other output of each pair --
      // lane l's register r pairs with lane 64 - l's register 15 - r --, instead of both lanes computing both.  Lane 0's
      // bins pair among themselves (64 r with 64 (16 - r)): its fetched and delivered registers are shifted by one and its
      // bin 512 is its own partner.
      cx<double> g[16], gp[8];
      auto pair_product = [&](const cx<double>& zu_k, const cx<double>& pu, const cx<double>& zn_k, const cx<double>& pn,
                              const cx<double>& w, cx<double>& at_k, cx<double>& at_partner) {
        cx<double> e, o, en, on;
        even_odd(zu_k, pu, e, o);
        even_odd(zn_k, pn, en, on);
        const cx<double> wo = cmul(w, o), won = cmul(w, on);
        const cx<double> U{e.re + wo.re, e.im + wo.im};
        const cx<double> Up{e.re - wo.re, -(e.im - wo.im)};                  // U[1024-k] = conj(E - w O)
        const cx<double> X{U.re + sign * (en.re + won.re), U.im + sign * (en.im + won.im)};
        const cx<double> Xp{Up.re + sign * (en.re - won.re), Up.im - sign * (en.im - won.im)};
        const cx<double> C{U.re * X.re + U.im * X.im, U.re * X.im - U.im * X.re};
        const cx<double> Cp{Up.re * Xp.re + Up.im * Xp.im, Up.re * Xp.im - Up.im * Xp.re};
        const cx<double> ec{C.re + Cp.re, C.im - Cp.im};
        const cx<double> d{C.re - Cp.re, C.im + Cp.im};
        const cx<double> oc{d.re * w.re + d.im * w.im, d.im * w.re - d.re * w.im};
        at_k = {ec.re - oc.im, -(ec.im + oc.re)};      // conj(8 Zc)
        at_partner = {ec.re + oc.im, ec.im - oc.re};
      };
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const cx<double> pu = partner_of(zu, r, c), pn = partner_of(zn, r, c);
        pair_product(zu[r], pu, zn[r], pn, c.post[64 * r], g[r], gp[r]);
      }
      cx<double> g512, unused;
      pair_product(zu[8], zu[8], zn[8], zn[8], c.post[64 * 8], g512, unused);     // lane 0's bin 512 (the other lanes' result is dropped)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        g[15 - j] = {__shfl(gp[j].re, c.partner), __shfl(gp[j].im, c.partner)};
        if (lane == 0) g[15 - j] = (j < 7) ? gp[j + 1] : g512;
      }
      // the second half's transform is the next frame's first-half transform
#pragma unroll
      for (int r = 0; r < 16; ++r) zu[r] = zn[r];
      // the frame's samples in the blocked layout (16 consecutive per lane) for the squared-difference terms: asked for
      // now, used behind the inverse transform
      Block16<TIn> qa, qb;
      if (kPrefetch) {
        __builtin_amdgcn_sched_barrier(0);
        qa.load(x + 16 * lane);
        qb.load(x + W + 16 * lane);
        __builtin_amdgcn_sched_barrier(0);
      }
      fft(g, c);
      // c[2m] = Re F[m] / 1024, c[2m+1] = -Im F[m] / 1024 (and the carried factor 8), m = lane + 64 r;
      // tau < 1024 <=> r < 8.
      // To the blocked layout through the plane.
      wave_lds_fence();
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int t = 2 * (64 * r + lane);
        c.plane_d[pad_slot(t)] = g[r].re * (1.0 / 8192.0);
        c.plane_d[pad_slot(t) + 1] = -g[r].im * (1.0 / 8192.0);
      }
      wave_lds_fence();
      double corr[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) corr[i] = c.plane_d[17 * lane + i];
      wave_lds_fence();

      // ---- squared-difference terms (pitchyinfast.c:96-117) ----
      double xa[16], xb[16];
      if (!kPrefetch) {
        qa.load(x + 16 * lane);
        qb.load(x + W + 16 * lane);
      }
      qa.get(xa);
      qb.get(xb);
      scale16<SCALED>(xa, sc);
      scale16<SCALED>(xb, sc);
      // the hop's own descriptors (silence flag, envelope, amplitude: hop_kernel's, SA:865-872, 1760-1804): a batch that
      // computes f0 does not launch a kernel and read the PCM from HBM once more for them; here, where the hop is in
      // registers in the blocked layout anyway
      if (a.hop_here) hop_descriptors(xa, hop_energy(xa), env_coef, env_pow, env_scan_of_lane[lane], a.amplitude, a.lay, rec, lane);
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        xa[i] *= xa[i];
        xb[i] *= xb[i];
        s0 += xa[i];
        s1 += xb[i];
      }
      s0 = wave_sum(s0);
      s1 = wave_sum(s1);
      // sqdiff[tau] = sum_{j<W} x[j+tau]^2 + sum_{j<W} x[j]^2 = 2 s0 + sum_{j<tau} (x[W+j]^2 - x[j]^2)
      double yin[16], run = 0.0;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        yin[i] = run;
        run += xb[i] - xa[i];
      }
      const double before = wave_scan_incl(run, lane) - run;
      // aubio's Ooura back end returns the inverse transform halved (spectral/fft.c:464-476: scale 1/N where
      // rdft needs 2/N), so "sqdiff - 2 r_t" is sqdiff - r_t in the reference's results
#pragma unroll
      for (int i = 0; i < 16; ++i) yin[i] = (2.0 * s0 + (before + yin[i])) - 2.0 * (0.5 * corr[i]);

      // ---- cumulative mean normalisation (pitchyinfast.c:146-153) ----
      double cum[16];
      run = 0.0;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (16 * lane + i > 0) run += yin[i];
        cum[i] = run;
      }
      const double cum_before = wave_scan_incl(run, lane) - run;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int tau = 16 * lane + i;
        const double tmp2 = cum_before + cum[i];
        // yin[tau] *= tau / tmp2 (pitchyinfast.c:148-150); tmp2 >= 0 is a sum of a^2 + b^2 - ab terms
        yin[i] = (tau == 0) ? 1.0 : ((tmp2 != 0) ? yin[i] * fast_div((double)tau, tmp2) : 1.0);
      }

      // ---- first dip below the tolerance, else the (last) global minimum (pitchyinfast.c:154-163) ----
      const double nxt = next_lane(yin[0], yin[0]);
      int first = kBig;
#pragma unroll
      for (int i = 15; i >= 0; --i) {
        const int p = 16 * lane + i;
        const double hi = (i == 15) ? nxt : yin[i + 1];
        if (p >= 2 && p + 3 < W && yin[i] < kTol && yin[i] < hi) first = p;
      }
      first = wave_min_i(first);
      int pos = first;
      if (first == kBig) {
        double m = yin[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) m = fmin(m, yin[i]);
        m = wave_min(m);
        int last = -1;
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (yin[i] == m) last = 16 * lane + i;
        pos = wave_max_i(last);
        if (pos < 0) pos = 0;
      }
      // fvec_quadratic_peak_pos (mathutils.c:494-506) and the confidence read back from the normalised buffer
      wave_lds_fence();
#pragma unroll
      for (int i = 0; i < 16; ++i) c.plane_d[17 * lane + i] = yin[i];
      wave_lds_fence();
      double period = (double)pos;
      if (pos != 0 && pos != W - 1) {
        const double y0 = c.plane_d[pad_slot(pos - 1)], y1 = c.plane_d[pad_slot(pos)], y2 = c.plane_d[pad_slot(pos + 1)];
        period = (double)pos + 0.5 * (y0 - y2) / (y0 - 2.0 * y1 + y2);
      }
      // o->peak_pos is a uint_t (pitchyinfast.c:38, 160-169); out-of-range casts are undefined there, clamped here
      int at = (period >= 0.0 && period < (double)W) ? (int)period : (period < 0.0 ? 0 : W - 1);
      if (!(period == period)) at = 0;
      const double conf_raw = 1.0 - c.plane_d[pad_slot(at)];
      wave_lds_fence();
      double f0 = (period > 0.0) ? (double)kSampleRate / (period + 0.) : 0.0;       // pitch.c:450-462
      if (au_silent(s0 + s1, kFft)) f0 = 0.0;                                 // pitch.c:399-406
      // SA:886-895: confidence / MMaxPitchConfidenceValue clipped to [0, 1]
      double conf = conf_raw / 0.25;
      conf = (conf < 0.0) ? 0.0 : (conf > 1.0 ? 1.0 : conf);
      if (!(conf == conf)) conf = 0.0;
      if (lane == 0) {
        rec[a.lay.f0] = nan_to_zero(f0);
        rec[a.lay.f0_conf] = conf;
        // hop silence (SA:865-868) parked in the fail-safe slot for afx_whiten.hip, which replaces it
        rec[a.lay.f0_safe] = au_silent(s0, kHop) ? 1.0 : 0.0;
      }
    }
  }
}

template <typename K>
hipError_t raise_lds_limit(K kernel, bool (&done)[16]) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 16 || !done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       kTimeLdsBytes);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 16) done[dev] = true;
  }
  return hipSuccess;
}

int time_grid(int n_chunks, int waves) {
  int cus = 256;
  int dev = 0;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  const int want = (n_chunks + waves - 1) / waves;
  return want < cus ? (want < 1 ? 1 : want) : cus;
}

}  // namespace

hipError_t launch_hop(const TimeArgs& a, hipStream_t stream) {
  if (a.n_chunks <= 0) return hipSuccess;
  const int grid = (a.n_chunks + 3) / 4 < 2048 ? (a.n_chunks + 3) / 4 : 2048;
  if (a.pcm_dtype == kPcmF32) hipLaunchKernelGGL((hop_kernel<float, false>), dim3(grid), dim3(256), 0, stream, a);
  else if (a.pcm_dtype == kPcmScaledF32) hipLaunchKernelGGL((hop_kernel<float, true>), dim3(grid), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((hop_kernel<double, false>), dim3(grid), dim3(256), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_acorr(const TimeArgs& a, hipStream_t stream) {
  if (a.n_chunks <= 0) return hipSuccess;
  static bool done_f[16] = {}, done_d[16] = {}, done_s[16] = {};
  const int grid = time_grid(a.n_chunks, kTimeWaves);
  hipError_t e;
  if (a.pcm_dtype == kPcmF32) {
    if ((e = raise_lds_limit(acorr_kernel<float, false>, done_f)) != hipSuccess) return e;
    hipLaunchKernelGGL((acorr_kernel<float, false>), dim3(grid), dim3(kTimeWaves * 64), kTimeLdsBytes, stream, a);
  } else if (a.pcm_dtype == kPcmScaledF32) {
    if ((e = raise_lds_limit(acorr_kernel<float, true>, done_s)) != hipSuccess) return e;
    hipLaunchKernelGGL((acorr_kernel<float, true>), dim3(grid), dim3(kTimeWaves * 64), kTimeLdsBytes, stream, a);
  } else {
    if ((e = raise_lds_limit(acorr_kernel<double, false>, done_d)) != hipSuccess) return e;
    hipLaunchKernelGGL((acorr_kernel<double, false>), dim3(grid), dim3(kTimeWaves * 64), kTimeLdsBytes, stream, a);
  }
  return hipGetLastError();
}

hipError_t launch_pitch(const TimeArgs& a, hipStream_t stream) {
  if (a.n_chunks <= 0) return hipSuccess;
  static bool done_f[16] = {}, done_d[16] = {}, done_s[16] = {};
  const int grid = time_grid(a.n_chunks, kTimeWaves);
  hipError_t e;
  if (a.pcm_dtype == kPcmF32) {
    if ((e = raise_lds_limit(pitch_kernel<float, false>, done_f)) != hipSuccess) return e;
    hipLaunchKernelGGL((pitch_kernel<float, false>), dim3(grid), dim3(kTimeWaves * 64), kTimeLdsBytes, stream, a);
  } else if (a.pcm_dtype == kPcmScaledF32) {
    if ((e = raise_lds_limit(pitch_kernel<float, true>, done_s)) != hipSuccess) return e;
    hipLaunchKernelGGL((pitch_kernel<float, true>), dim3(grid), dim3(kTimeWaves * 64), kTimeLdsBytes, stream, a);
  } else {
    if ((e = raise_lds_limit(pitch_kernel<double, false>, done_d)) != hipSuccess) return e;
    hipLaunchKernelGGL((pitch_kernel<double, false>), dim3(grid), dim3(kTimeWaves * 64), kTimeLdsBytes, stream, a);
  }
  return hipGetLastError();
}

}  // namespace afx
