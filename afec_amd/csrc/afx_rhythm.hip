// afec_amd/csrc/afx_rhythm.hip -- the rhythm tracker behind the per-frame loop: the second, 512/128
// loop of TSampleAnalyser::AnalyzeLowLevelDescriptors (SampleAnalyser.cpp:983-1048).
//   TRhythmTracker                       Source/Crawler/FeatureExtraction/Source/RhythmTracker.cpp   ("RT.cpp")
//   TOnsetFftProcessor / TOnsetDetector  Source/Core/AudioTypes/Source/OnsetDetector.cpp             ("OD.cpp")
//   TCannyWindow                         Source/Crawler/FeatureExtraction/Source/CannyWindow.cpp     ("CW.cpp")
//   aubio beat tracking                  3rdParty/Aubio/Dist/src/tempo/beattracking.c ("bt.c"), mathutils.c
//
// Kernels (256 threads per workgroup; one workgroup per file, the post kernel one per file and onset function):
//   onset_function_kernel  rounds of 16 frames: (1) one 16-lane group per frame computes the 512-point real FFT as a
//                          256-point complex FFT (16 x 16 in registers, one LDS exchange) + untangle, magnitude and phase
//                          of bins 0..254 as float; (2) thread = bin walks the 16 frames in order: adaptive-max whitening
//                          follower (double state per bin), the power term and the rectified-complex deviation of the bin;
//                          (3) one lane per (frame, function) adds the 255 terms in the reference's order.  Nothing but the
//                          two onset-function values per frame leaves the CU.
//   rhythm_post_kernel     per file and function: sliding median removal + detection, Canny sharpening, autocorrelation +
//                          comb filterbank + Rayleigh weighting of the beat tracker, peaks, strength, contrast, and the
//                          duration heuristics of the final tempo.  Sums the reference accumulates serially are
//                          accumulated serially here (staged through LDS, one lane adds): their value depends on the order.
//
// Float semantics: the reference keeps the polar spectrum, the whitened magnitudes and both onset functions in `float`;
// every expression below has the type of its counterpart and the file is compiled without FMA contraction.  The only
// operations that are not the reference's bit for bit are libm calls: atan2 / sqrt of FFT outputs that differ in the
// last bits, pow / log of the contrast (cosf is glibc's algorithm, cosf_glibc below).
#include "afx_internal.h"

#include "afx_device.h"
#include "afx_fft32.h"

#pragma clang fp contract(off)

// waves per SIMD the kernels are compiled for (A/B on MI355X, many-file workloads: 2 / 3 -> 3 / 6 waves: -18 % time;
// the onset kernel then holds 168 VGPRs with 32 B of scratch, the post kernel 80 VGPRs with 384 B -- its VALU pipe is
// only ~15 % busy, what it needs is more workgroups in flight to hide the barrier and memory latencies)
constexpr int kOnsetWavesPerSimd = 3, kPostWavesPerSimd = 6;

namespace afx {
namespace {

constexpr int kRtHop = 128, kRtBins = 255;   // TempoHopSize (SampleAnalyser.cpp:986); mNumbins = 512/2 - 1 (OD.cpp:43)
constexpr int kRound = 16;                   // frames per round
constexpr int kRow = 257;                    // floats per row of the polar / term planes (bank spread)
constexpr int kPlane = 272;                  // doubles per exchange plane of a 16-lane group (16 x 17)

using C = f32x32::cx<double>;

__device__ __forceinline__ float phase_rewrap(float p) {   // SPhaseRewrap, OD.cpp:18-22
  constexpr float pi = (float)3.1415926535897932384626433832795, two_pi = (float)6.2831853071795864769252867665590,
                  inv = (float)0.15915494309189533576888376337251;
  return (p > -pi && p < pi) ? p : p + two_pi * (1.f + floorf((-pi - p) * inv));
}

// cosf as glibc (>= 2.28) computes it, the libm the reference calls on Linux (::cosf, OD.cpp:441): double-precision
// reduction by pi/2 and the sincosf polynomials of sysdeps/ieee754/flt-32/s_sincosf.h (coefficients of its
// __sincosf_table; the ARM optimized-routines algorithm), rounded to float once.  Valid for |y| < 120 -- the argument is
// a rewrapped phase.  With the products fused as below it equals this container's cosf on 10^8 arguments of [-pi, pi]
// (a CPU restatement, fused and unfused, against libm: 0 differences).
__device__ __forceinline__ float cosf_glibc(float y) {
  const unsigned top = (__float_as_uint(y) >> 20) & 0x7ffu;
  double x = (double)y;
  int n = 0;
  if (top >= 0x3f4u) {                                             // |y| >= pi/4
    const double r = x * 0x1.45F306DC9C883p+23;                    // 2/pi * 2^24
    n = ((int)r + 0x800000) >> 24;
    x = __builtin_fma(-(double)n, 0x1.921FB54442D18p0, x);
  }
  const double x2 = x * x;
  const double xs = (((n + 1) & 2) != 0) ? -x : x;                 // sign[n & 3] = {1, -1, -1, 1}
  const double x3 = xs * x2, t1 = __builtin_fma(x2, -0x1.994eb3774cf24p-13, 0x1.1107605230bc4p-7), x7 = x3 * x2;
  const double sp = __builtin_fma(x7, t1, __builtin_fma(x3, -0x1.555545995a603p-3, xs));
  const double x4 = x2 * x2, d2 = __builtin_fma(x2, 0x1.99343027bf8c3p-16, -0x1.6c087e89a359dp-10);
  const double d1 = __builtin_fma(x2, -0x1.ffffffd0c621cp-2, 1.0), x6 = x4 * x2;
  const double cp = __builtin_fma(x6, d2, __builtin_fma(x4, 0x1.55553e1068f19p-5, d1));
  const double res = (n & 1) ? sp : ((n & 2) ? -cp : cp);          // quadrant n: cos, -sin, -cos, sin of the reduced angle
  return (top < 0x398u) ? 1.0f : (float)res;                       // |y| < 2^-12
}

__device__ __forceinline__ void load2(const float* p, double& a, double& b) {
  const float2 v = *reinterpret_cast<const float2*>(p);
  a = (double)v.x; b = (double)v.y;
}
__device__ __forceinline__ void load2(const double* p, double& a, double& b) {
  const double2 v = *reinterpret_cast<const double2*>(p);
  a = v.x; b = v.y;
}

// atan2 and sqrt of the polar spectrum.  TAudioMath::Phase / Magnitude are the double libm calls rounded to float
// (AudioMath.cpp:497-503, 637-643).  The device library's double atan2 is 104 instructions (division with scaling,
// a degree-19 polynomial, inf / nan handling) and was 40 % of the onset kernel; this one is 43: the ratio of the smaller to
// the larger magnitude is reduced against the nearest c = k / 16, t = (mn - c mx) / (mx + c mn) with |t| <= 1 / 32, atan t
// as five Taylor terms (the next one is 7e-20 relative), atan c from a table, then the octant.  2 ulp of a double at
// worst; rounded to float it equalled glibc's atan2 on 60 000 000 arguments of mixed magnitudes, octants and signed
// zeros (a CPU restatement with the same operations).  Inputs are finite (FFT outputs).
__device__ const double kAtanOfSixteenths[17] = {
    0x0.0p+0, 0x1.ff55bb72cfdeap-5, 0x1.fd5ba9aac2f6ep-4, 0x1.7b97b4bce5b02p-3, 0x1.f5b75f92c80ddp-3, 0x1.362773707ebccp-2,
    0x1.6f61941e4def1p-2, 0x1.a64eec3cc23fdp-2, 0x1.dac670561bb4fp-2, 0x1.0657e94db30d0p-1, 0x1.1e00babdefeb4p-1,
    0x1.345f01cce37bbp-1, 0x1.4978fa3269ee1p-1, 0x1.5d58987169b18p-1, 0x1.700a7c5784634p-1, 0x1.819d0b7158a4dp-1,
    0x1.921fb54442d18p-1};

__device__ __forceinline__ double polar_atan2(double y, double x) {
  const double ax = fabs(x), ay = fabs(y);
  const double mx = fmax(ax, ay), mn = fmin(ax, ay);
  const double xa = mn * __builtin_amdgcn_rcp(mx);          // 0 / 0: nan, converted to k = 0
  int k = (int)fma(xa, 16.0, 0.5);
  k = k > 16 ? 16 : k;
  const double c = (double)k * 0.0625;
  const double num = fma(-c, mx, mn);
  double den = fma(c, mn, mx);
  den = (mx == 0.0) ? 1.0 : den;
  double r = __builtin_amdgcn_rcp(den);
  double e = fma(-den, r, 1.0);
  r = fma(e, r, r);
  e = fma(-den, r, 1.0);
  r = fma(e, r, r);
  double t = num * r;
  e = fma(-den, t, num);
  t = fma(e, r, t);
  const double t2 = t * t;
  double p = -1.0 / 11.0;
  p = fma(p, t2, 1.0 / 9.0);
  p = fma(p, t2, -1.0 / 7.0);
  p = fma(p, t2, 1.0 / 5.0);
  p = fma(p, t2, -1.0 / 3.0);
  const double s = (t * t2) * p;
  double a = kAtanOfSixteenths[k] + (t + s);
  if (ay > ax) a = (0x1.921fb54442d18p+0 - a) + 0x1.1a62633145c07p-54;     // pi / 2 in two parts
  if (__builtin_signbit(x)) a = (0x1.921fb54442d18p+1 - a) + 0x1.1a62633145c07p-53;
  return __builtin_copysign(a, y);
}

// sqrt(s), s >= 0 finite and far from the ends of the exponent range: reciprocal square root + two coupled Newton steps
// (the device library's sequence without its rescaling and special cases)
__device__ __forceinline__ double polar_sqrt(double s) {
  const double r0 = __builtin_amdgcn_rsq(s);
  double g = s * r0, h = 0.5 * r0;
  double e = fma(-h, g, 0.5);
  g = fma(g, e, g);
  h = fma(h, e, h);
  e = fma(-g, g, s);
  g = fma(e, h, g);
  return (s == 0.0) ? 0.0 : g;
}

// natural-order 16-point forward DFT of v (first radix-4 stage + the rest, afx_fft32.h)
__device__ __forceinline__ void dft16(C (&v)[16]) {
#pragma unroll
  for (int b = 0; b < 4; ++b) f32x32::radix4(v[b], v[b + 4], v[b + 8], v[b + 12]);
  f32x32::dft16_rest(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15]);
}

// TOnsetFftProcessor::LoadFrame (OD.cpp:116-160) for one frame by one 16-lane group (lane q): window, FFT, magnitude and
// phase of bins q + 16 r as float.  z[j] = (w x)[2j] + i (w x)[2j+1], j = 16 n1 + n2: lane n2 = q holds n1 = 0..15.  The
// window table carries the 1/2 of the real-input untangle (an exact scaling).  xg: the group's exchange plane (kPlane
// doubles of LDS).
// SCALED (kPcmScaledF32): the arena holds LoadSample's float signal, `scale` is the buffer's FinalScaling and the
// reference's sample is their product (pcm_double, afx_device.h).
template <typename PCM, bool SCALED>
__device__ __forceinline__ void onset_polar_frame(const PCM* xf, double scale, double* xg, const RhythmArgs& a, int q,
                                                  float (&magf)[16], float (&phf)[16]) {
  const C* tw = reinterpret_cast<const C*>(a.tw256);   // [n2][k1]: w256^(n2 k1)
  const C* ut = reinterpret_cast<const C*>(a.ut512);   // [r][q]:   w512^(q + 16 r)
  C v[16];
#pragma unroll
  for (int n1 = 0; n1 < 16; ++n1) {
    const int j = 16 * n1 + q;
    double x0, x1, w0, w1;
    load2(xf + 2 * j, x0, x1);
    if (SCALED) {
      x0 = pcm_double<true>(x0, scale);
      x1 = pcm_double<true>(x1, scale);
    }
    load2(a.window + 2 * j, w0, w1);
    v[n1] = {w0 * x0, w1 * x1};
  }
  dft16(v);                                         // Y[k1][n2 = q]
  C u[16];
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) xg[k1 * 17 + q] = v[k1].re;
  wave_lds_fence();
#pragma unroll
  for (int n2 = 0; n2 < 16; ++n2) u[n2].re = xg[q * 17 + n2];
  wave_lds_fence();
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) xg[k1 * 17 + q] = v[k1].im;
  wave_lds_fence();
#pragma unroll
  for (int n2 = 0; n2 < 16; ++n2) u[n2].im = xg[q * 17 + n2];
  wave_lds_fence();
  // lane k1 = q: factors w256^(n2 k1) fused into the first radix-4 stage of the DFT over n2
  {
    C w[16];
#pragma unroll
    for (int n2 = 1; n2 < 16; ++n2) w[n2] = tw[n2 * 16 + q];
    f32x32::radix4_tw3(u[0], u[4], u[8], u[12], w[4], w[8], w[12]);
#pragma unroll
    for (int b = 1; b < 4; ++b) f32x32::radix4_tw4(u[b], u[b + 4], u[b + 8], u[b + 12], w[b], w[b + 4], w[b + 8], w[b + 12]);
  }
  f32x32::dft16_rest(u[0], u[1], u[2], u[3], u[4], u[5], u[6], u[7], u[8], u[9], u[10], u[11], u[12], u[13], u[14], u[15]);
  // u[r] = Z[k], k = q + 16 r.  Untangle with the partner Z[(256 - k) & 255] through the group's plane.
  C p[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) xg[q + 16 * r] = u[r].re;
  wave_lds_fence();
#pragma unroll
  for (int r = 0; r < 16; ++r) p[r].re = xg[(256 - (q + 16 * r)) & 255];
  wave_lds_fence();
#pragma unroll
  for (int r = 0; r < 16; ++r) xg[q + 16 * r] = u[r].im;
  wave_lds_fence();
#pragma unroll
  for (int r = 0; r < 16; ++r) p[r].im = xg[(256 - (q + 16 * r)) & 255];
  // kPolarGroup bins at a time (scheduling barriers in between): all sixteen interleaved keep more values alive than
  // there are registers at three waves per SIMD (116 B of scratch).  onset_function_kernel on the C4 share: device library's
  // atan2 / sqrt 8.60 ms; these with groups of 16 / 8 / 4 / 2 bins: 8.31 / 7.90 / 7.88 / 7.73 ms (tools/ab_rhythm.sh)
  constexpr int kPolarGroup = 2;
#pragma unroll
  for (int r0 = 0; r0 < 16; r0 += kPolarGroup) {
#pragma unroll
    for (int r = r0; r < r0 + kPolarGroup; ++r) {
      const C A = u[r], B = p[r], w = ut[r * 16 + q];
      const double er = A.re + B.re, ei = A.im - B.im, orr = A.im + B.im, oi = B.re - A.re;
      const double re = er + (w.re * orr - w.im * oi);
      // the reference transform is e^{+i} (ooura_cdft(.., 1, ..), Fourier.cpp:243-262): conjugate of this one
      double im = -(ei + (w.re * oi + w.im * orr));
      if (q + 16 * r == 0) im = 0.0;                          // Im[0] of a real frame is +0
      magf[r] = (float)polar_sqrt(re * re + im * im);         // TAudioMath::Magnitude, AudioMath.cpp:497-503
      phf[r] = (float)polar_atan2(im, re);                    // TAudioMath::Phase, AudioMath.cpp:637-643
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Whiten (OD.cpp:186-240), the follower step of one bin: psp is the adaptive maximum (double), the return value the
// float the magnitude is divided by
__device__ __forceinline__ float onset_follow(float mg, double& psp, float relax_coef) {
  double value = (double)fabsf(mg);
  const double old = psp;
  if (value < old) value = value + (old - value) * (double)relax_coef;
  psp = value;
  const double fl = (double)0.1f;                                     // mWhiteningFloor, OD.cpp:47
  return (float)((fl > psp) ? fl : psp);
}

// sqrtf, correctly rounded, for arguments that are neither tiny, huge, negative nor NaN: v_sqrt_f32 (1 ulp) and the choice
// among its neighbours by the sign of the residuals -- the compiler's own expansion without its rescaling of denormal
// arguments and its special cases (18 -> 9 instructions).  The one caller's argument is a^2 + b^2 - a b c with
// 0.01 < b <= 1, 0 <= a <= 1 (whitened magnitudes: the follower is never below the magnitude it divides), |c| <= 1:
// between 7e-5 and 3.
__device__ __forceinline__ float sqrtf_midrange(float x) {
  const float s = __builtin_amdgcn_sqrtf(x);
  const float below = __int_as_float(__float_as_int(s) - 1), above = __int_as_float(__float_as_int(s) + 1);
  const float r_below = __builtin_fmaf(-below, s, x), r_above = __builtin_fmaf(-above, s, x);
  float r = (r_below <= 0.f) ? below : s;
  r = (r_above > 0.f) ? above : r;
  return r;
}

// The terms of both onset functions for one bin of one frame (OD.cpp:380-458) from the whitened magnitude m and the
// phase ph of the frame and what the frame before left: pred_mag = its |m|, yester_phase = its phase, yester_diff =
// its rewrapped phase step.  Returns the rectified complex-domain deviation, pw the power term.
__device__ __forceinline__ float onset_terms(float m, float ph, float pred_mag, float yester_phase, float yester_diff, float& pw) {
  const float cur = fabsf(m);
  float dev = 0.f;
  if (cur > 0.01f) {                                                  // mOdfparam, OD.cpp:299
    if (!(cur < pred_mag)) {                                          // Rectify: ignore decreasing bins
      const float pred_phase = yester_phase + yester_diff;
      float d = pred_phase - ph;
      d = phase_rewrap(d);
      const float cs = cosf_glibc(d);
      dev = sqrtf_midrange(pred_mag * pred_mag + cur * cur - pred_mag * cur * cs);
    }
  }
  pw = m * m;
  return dev;
}

// the two onset-function values of frame `fr` of a round from its 255 + 1 staged terms (rows of kRow floats), added in the
// reference's order (15 at a time in registers so that the LDS reads of a group are in flight together)
__device__ __forceinline__ float onset_sum(const float* plane_p, const float* plane_c, int fr, int type, const RhythmArgs& a) {
  float v;
  if (type == 0) {          // kFunctionRComplex: double sum of the float deviations, OD.cpp:398-458
    const float* row = plane_c + fr * kRow;
    double total = 0.0;
    for (int i0 = 0; i0 < kRtBins; i0 += 15) {
      float t[15];
#pragma unroll
      for (int k = 0; k < 15; ++k) t[k] = row[i0 + k];
#pragma unroll
      for (int k = 0; k < 15; ++k) total += (double)t[k];
    }
    v = (float)total;
    v *= a.norm_complex;
  } else {                  // kFunctionPower: float sum, OD.cpp:380-388 (mNyquist = Im[0] = 0)
    const float* row = plane_p + fr * kRow;
    v = (0.f * 0.f) + row[255];
    for (int i0 = 0; i0 < kRtBins; i0 += 15) {
      float t[15];
#pragma unroll
      for (int k = 0; k < 15; ++k) t[k] = row[i0 + k];
#pragma unroll
      for (int k = 0; k < 15; ++k) v += t[k];
    }
    v *= a.norm_power;
  }
  return v;
}

template <typename PCM, bool SCALED>
__global__ __launch_bounds__(256, kOnsetWavesPerSimd) void onset_function_kernel(RhythmArgs a) {
  __shared__ double s_x[kRound * kPlane];   // 34,816 B: exchange planes of the FFT stage, then the float polar / term rows
  const RhythmFile f = a.files[blockIdx.x];
  const int T = f.frames;
  if (T <= 0 || f.long_slot > 0) return;    // long files of small batches: the three kernels below
  const int tid = threadIdx.x, g = tid >> 4, q = tid & 15;
  const PCM* x = reinterpret_cast<const PCM*>(a.pcm) + f.sample_off;
  double* xg = s_x + g * kPlane;
  float* plane_p = reinterpret_cast<float*>(s_x);      // [16][257]: magnitudes, then the power terms m^2
  float* plane_c = plane_p + kRound * kRow;            // [16][257]: phases, then the complex-domain deviations
  // state of bin `tid` across the frames of the file
  double psp = 0.0;                         // whitening follower, OD.cpp:63-65, 186-230
  float pred_mag = 0.f, yester_phase = 0.f, yester_diff = 0.f;   // mpOther, OD.cpp:300-305

  for (int t0 = 0; t0 < T; t0 += kRound) {
    const int nf = min(kRound, T - t0);
    float magf[16], phf[16];
    if (g < nf) onset_polar_frame<PCM, SCALED>(x + (int64_t)(t0 + g) * kRtHop, f.scale, xg, a, q, magf, phf);
    __syncthreads();   // every group is done with its exchange plane: the polar rows take the space
    if (g < nf) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        plane_p[g * kRow + q + 16 * r] = magf[r];
        plane_c[g * kRow + q + 16 * r] = phf[r];
      }
    }
    __syncthreads();
    // ---- thread = bin: Whiten (OD.cpp:186-240) + the terms of both onset functions (OD.cpp:380-458), frame by frame ----
    if (tid < kRtBins) {
      // (the next frame's pair is asked for before this frame's terms are computed: its row is not written until then)
      float mg_next = plane_p[tid], ph_next = plane_c[tid];
      for (int fr = 0; fr < nf; ++fr) {
        const float mg = mg_next, ph = ph_next;
        const int nx = (fr + 1 < nf) ? fr + 1 : fr;
        mg_next = plane_p[nx * kRow + tid];
        ph_next = plane_c[nx * kRow + tid];
        const float m = mg / onset_follow(mg, psp, a.relax_coef);
        float pw;
        const float dev = onset_terms(m, ph, pred_mag, yester_phase, yester_diff, pw);
        pred_mag = fabsf(m);
        const float diff = ph - yester_phase;
        yester_phase = ph;
        yester_diff = phase_rewrap(diff);
        plane_p[fr * kRow + tid] = pw;
        plane_c[fr * kRow + tid] = dev;
        // mDC is (float)Re[0] and mBin[0].mMagn is |Re[0]|: the same float up to sign, whitened by the same follower
        // values (psp[0] and psp[1] see the same input), so mDC^2 is the power term of bin 0
        if (tid == 0) plane_p[fr * kRow + 255] = pw;
      }
    }
    __syncthreads();
    // ---- one lane per (frame, function): the sums in the reference's order; the two functions in different waves, so
    //      that their dependent-add chains (255 double adds, 255 float adds) run side by side instead of as the two
    //      sides of a divergent branch ----
    if ((tid & 63) < nf && tid < 128) {
      const int fr = tid & 63, type = tid >> 6;
      a.odf[(f.frame0 + t0 + fr) * 2 + type] = onset_sum(plane_p, plane_c, fr, type, a);
    }
    __syncthreads();
  }
}

// ---- long files of a small batch (RhythmFile::long_slot > 0): the same arithmetic as three kernels, so that a lone
//      20 s file is not 430 dependent rounds of one workgroup.  Only the follower is a recurrence over the frames; the
//      deviation needs the two frames before it, nothing older.
//        onset_polar_kernel   one workgroup per round of 16 frames: FFT -> (magnitude, phase) floats in global memory
//        onset_follow_kernel  one workgroup per file, thread = bin, frames in order: the adaptive maximum, i.e. what
//                             the magnitude is divided by (long_den)
//        onset_terms_kernel   one workgroup per round: the terms of both functions (they need frames t, t-1, t-2) and the
//                             ordered sums
//      polar: [long frames][256] float2 and den: [long frames][256] float, the long files back to back
//      (RhythmArgs::long_frame_off) ----
__device__ __forceinline__ int long_file_of_round(const RhythmArgs& a, int round, int& round_in_file) {
  int i = 0;
  while (i + 1 < a.n_long && round >= a.long_round_off[i + 1]) ++i;
  round_in_file = round - a.long_round_off[i];
  return i;
}

template <typename PCM, bool SCALED>
__global__ __launch_bounds__(256, kOnsetWavesPerSimd) void onset_polar_kernel(RhythmArgs a) {
  __shared__ double s_x[kRound * kPlane];
  int round;
  const int slot = long_file_of_round(a, blockIdx.x, round);
  const RhythmFile f = a.files[a.long_files[slot]];
  const int T = f.frames, t0 = round * kRound, tid = threadIdx.x, g = tid >> 4, q = tid & 15;
  if (t0 + g >= T) return;
  const PCM* x = reinterpret_cast<const PCM*>(a.pcm) + f.sample_off;
  float magf[16], phf[16];
  onset_polar_frame<PCM, SCALED>(x + (int64_t)(t0 + g) * kRtHop, f.scale, s_x + g * kPlane, a, q, magf, phf);
  float2* row = a.long_polar + (a.long_frame_off[slot] + t0 + g) * 256;
#pragma unroll
  for (int r = 0; r < 16; ++r) row[q + 16 * r] = float2{magf[r], phf[r]};
}

__global__ __launch_bounds__(256) void onset_follow_kernel(RhythmArgs a) {
  const int slot = blockIdx.x, tid = threadIdx.x;
  const int T = a.files[a.long_files[slot]].frames;
  const float2* p = a.long_polar + a.long_frame_off[slot] * 256 + tid;
  float* den = a.long_den + a.long_frame_off[slot] * 256 + tid;
  double psp = 0.0;
  // The chain per frame is a handful of dependent operations; a global load is a thousand cycles.  Three batches of 16
  // frames rotate: while one is worked on, the loads of the next two are in flight.  A file's rows are padded to a
  // multiple of the three batches (kRhythmLongPad): no bounds checks, the rows behind the last frame are scratch.
  // Only the recurrence runs here; the division by the follower is onset_terms_kernel's.
  constexpr int B = kRhythmLongPad / 3;
  float m0[B], m1[B], m2[B];
  auto load = [&](float (&m)[B], const float2* from) {
#pragma unroll
    for (int k = 0; k < B; ++k) m[k] = from[k * 256].x;
  };
  auto work = [&](float (&m)[B], float* to) {
#pragma unroll
    for (int k = 0; k < B; ++k) to[k * 256] = onset_follow(m[k], psp, a.relax_coef);
  };
  const int steps = (T + kRhythmLongPad - 1) / kRhythmLongPad;
  load(m0, p);
  load(m1, p + B * 256);
  load(m2, p + 2 * B * 256);
  for (int i = 0; i < steps; ++i) {
    const bool more = i + 1 < steps;     // (the loads behind the last step would leave the file's rows)
    const float2* const nxt = p + (more ? 3 * B * 256 : 0);
    work(m0, den);
    load(m0, nxt);
    work(m1, den + B * 256);
    load(m1, nxt + B * 256);
    work(m2, den + 2 * B * 256);
    load(m2, nxt + 2 * B * 256);
    p = nxt;
    den += 3 * B * 256;
  }
}

__global__ __launch_bounds__(256, kOnsetWavesPerSimd) void onset_terms_kernel(RhythmArgs a) {
  __shared__ float s_p[kRound * kRow], s_c[kRound * kRow];
  int round;
  const int slot = long_file_of_round(a, blockIdx.x, round);
  const RhythmFile f = a.files[a.long_files[slot]];
  const int T = f.frames, t0 = round * kRound, tid = threadIdx.x;
  const int nf = min(kRound, T - t0);
  const float2* base = a.long_polar + a.long_frame_off[slot] * 256 + tid;
  if (tid < kRtBins) {
    // (whitened magnitude, phase) of frames t0 - 2 .. t0 + nf - 1; the frames before the file are zeros (mpOther, OD.cpp:300-305)
    const float* const den = a.long_den + a.long_frame_off[slot] * 256 + tid;
    auto at = [&](int t) -> float2 {
      if (t < 0) return float2{0.f, 0.f};
      const float2 v = base[(int64_t)t * 256];
      return float2{v.x / den[(int64_t)t * 256], v.y};
    };
    const float2 v2 = at(t0 - 2), v1 = at(t0 - 1);
    float pred_mag = fabsf(v1.x), yester_phase = v1.y, yester_diff = phase_rewrap(v1.y - v2.y);
    if (t0 == 0) yester_diff = 0.f;            // (rewrap(0 - 0) is 0 as well: kept explicit)
    for (int fr = 0; fr < nf; ++fr) {
      const float2 v = at(t0 + fr);
      float pw;
      const float dev = onset_terms(v.x, v.y, pred_mag, yester_phase, yester_diff, pw);
      pred_mag = fabsf(v.x);
      const float diff = v.y - yester_phase;
      yester_phase = v.y;
      yester_diff = phase_rewrap(diff);
      s_p[fr * kRow + tid] = pw;
      s_c[fr * kRow + tid] = dev;
      if (tid == 0) s_p[fr * kRow + 255] = pw;
    }
  }
  __syncthreads();
  if ((tid & 63) < nf && tid < 128) {
    const int fr = tid & 63, type = tid >> 6;
    a.odf[(f.frame0 + t0 + fr) * 2 + type] = onset_sum(s_p, s_c, fr, type, a);
  }
}

// ---- post kernel helpers ----------------------------------------------------------------------------------------

constexpr int kStage = 1024;
constexpr int kMaxLdsFrames = kRhythmLdsFrames;

// sum of gen(0) .. gen(n-1) accumulated in index order (TStatistics::Sum, Statistics.cpp:236-245; fvec_sum): the terms are
// produced by all threads, one lane adds them.  Every thread gets the result.
template <typename Gen>
__device__ double serial_sum(int n, Gen gen, double* s_stage, double* s_result) {
  double acc = 0.0;
  for (int base = 0; base < n; base += kStage) {
    const int m = min(kStage, n - base);
    for (int i = threadIdx.x; i < m; i += blockDim.x) s_stage[i] = gen(base + i);
    __syncthreads();
    if (threadIdx.x == 0) {
      int i = 0;
      for (; i + 8 <= m; i += 8) {       // eight values in flight, added in order
        const double2 p0 = *reinterpret_cast<const double2*>(s_stage + i), p1 = *reinterpret_cast<const double2*>(s_stage + i + 2),
                      p2 = *reinterpret_cast<const double2*>(s_stage + i + 4), p3 = *reinterpret_cast<const double2*>(s_stage + i + 6);
        acc += p0.x; acc += p0.y; acc += p1.x; acc += p1.y; acc += p2.x; acc += p2.y; acc += p3.x; acc += p3.y;
      }
      for (; i < m; ++i) acc += s_stage[i];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *s_result = acc;
  __syncthreads();
  const double r = *s_result;
  __syncthreads();
  return r;
}

// TStatistics::Mean (Statistics.cpp:249-266) from the in-order sum
__device__ __forceinline__ double mean_of(double sum, int n, double first) {
  return (n >= 2) ? sum / (double)n : ((n == 1) ? first : 0.0);
}

// no value inside +-24 frames exceeds x[i] (RT.cpp:362-372, 643-653)
__device__ __forceinline__ bool window_peak(const double* x, int n, int i) {
  const double c = x[i];
  const int lo = max(0, i - 24), hi = min(n - 1, i + 24);
  bool ok = true;
  for (int j = lo; j <= hi; ++j) ok = ok && !(x[j] > c);
  return ok;
}

// aubio_autocorr (mathutils.c:652-666): acf[i] = (sum_m S[m] S[m+i]) / (T - i), every lag accumulated in the order of m.
// A thread owns four consecutive lags and slides a four-value register window over S, so that four multiply-adds cost
// one broadcast read of S[m] and one read of S[m+i+3]; with the series staged in LDS as four planes (S[4q+r] at plane
// r, slot q) the second read is conflict-free.
template <bool STAGED>
__device__ __forceinline__ void autocorrelation(const double* S, const double* lds, int plane, int T, double* acf) {
  auto at = [&](int j) -> double { return STAGED ? lds[(j & 3) * plane + (j >> 2)] : S[j]; };
  for (int base = 4 * (int)threadIdx.x; base < T; base += 4 * 256) {
    double a0 = 0., a1 = 0., a2 = 0., a3 = 0.;
    const int n3 = T - base - 3;                 // terms of lag base + 3
    int m = 0;
    if (n3 > 0) {
      double x0 = at(base), x1 = at(base + 1), x2 = at(base + 2);
#pragma unroll 4
      for (; m < n3; ++m) {
        const double sm = at(m), x3 = at(m + base + 3);
        a0 += sm * x0; a1 += sm * x1; a2 += sm * x2; a3 += sm * x3;
        x0 = x1; x1 = x2; x2 = x3;
      }
    }
    for (; m + base < T; ++m) {                  // the up to three last terms of the shorter lags
      const double sm = at(m);
      a0 += sm * at(m + base);
      if (m + base + 1 < T) a1 += sm * at(m + base + 1);
      if (m + base + 2 < T) a2 += sm * at(m + base + 2);
    }
    acf[base] = a0 / (double)(T - base);
    if (base + 1 < T) acf[base + 1] = a1 / (double)(T - base - 1);
    if (base + 2 < T) acf[base + 2] = a2 / (double)(T - base - 2);
    if (base + 3 < T) acf[base + 3] = a3 / (double)(T - base - 3);
  }
}

__device__ __forceinline__ unsigned long long order_key(double v) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_value(unsigned long long k) {
  const unsigned long long u = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
  return __longlong_as_double((long long)u);
}

__global__ __launch_bounds__(256, kPostWavesPerSimd) void rhythm_post_kernel(RhythmArgs a) {
  __shared__ __align__(16) double s_stage[kStage];
  __shared__ double s_result[2];
  __shared__ float s_win[520];
  extern __shared__ double s_series[];     // the sharpened onsets of the file, de-interleaved by 4, when they fit
  __shared__ float s_post[257];
  __shared__ float s_vlo[256], s_vhi[256];
  __shared__ unsigned long long s_mask[4];
  __shared__ int s_wsum[4];
  __shared__ unsigned s_flagbits[kStage / 32];
  __shared__ unsigned s_hist[256];
  __shared__ int s_int[4];
  __shared__ unsigned long long s_key;
  // one workgroup per (file, onset function): the two chains of a file are independent up to the final tempo, which
  // rhythm_final_kernel derives from both (a lone long file is latency-bound: two workgroups halve its time)
  const RhythmFile f = a.files[blockIdx.x >> 1];
  const int T = f.frames, tid = threadIdx.x;
  double* out = a.scalars + (int64_t)(blockIdx.x >> 1) * 14;
  if (T <= 0) {
    if (tid < 6) out[6 * (blockIdx.x & 1) + tid] = 0.0;
    if (tid == 0) out[12 + (blockIdx.x & 1)] = 0.0;
    return;
  }
  const int64_t tot = a.total_frames;
  const int rate = a.sample_rate;
  double tempo[2] = {0.0, 0.0}, conf[2] = {0.0, 0.0};
  int last_loud[2] = {0, 0};

  {
    const int type = blockIdx.x & 1;
    double* raw = a.scratch + (int64_t)(0 + type) * tot + f.frame0;   // TRhythmTracker::Onsets
    double* S = a.scratch + (int64_t)(2 + type) * tot + f.frame0;     // SharpenedOnsets
    double* W = a.scratch + (int64_t)(4 + type) * tot + f.frame0;     // convolution, then the autocorrelation, then flags
    double* AO = a.scratch + (int64_t)(6 + type) * tot + f.frame0;    // comb filterbank output [T / 4]
    double* o6 = out + 6 * type;
    const float* odf = a.odf + f.frame0 * 2 + type;
    const float thresh = a.thresh[type];
    const int med = a.medspan, mingap = a.mingap[type];

    // ---- DetectOnset (OD.cpp:549-587): median of the last `med` onset-function values removed, then the gap rule ----
    if (tid == 0) { s_post[256] = 0.f; s_int[0] = 0; s_int[1] = 0; }   // mOdfvalpost = 0, mGapLeft = 0, onset count
    __syncthreads();
    for (int c0 = 0; c0 < T; c0 += 256) {
      const int m = min(256, T - c0);
      for (int i = tid; i < 256 + med; i += 256) {
        const int fr = c0 - (med - 1) + i;
        s_win[i] = (fr >= 0 && fr < T) ? odf[(int64_t)fr * 2] : 0.f;
      }
      __syncthreads();
      // Median of every span by rank: element e of the staged values is the rank-r value of a span when fewer than
      // r + 1 values of the span lie below it and more than r do not exceed it.  Its rank is counted once for the first
      // span that holds it and updated by the value that leaves and the one that enters for the next ones.
      const int r_hi = med >> 1, r_lo = (med & 1) ? r_hi : r_hi - 1;   // sorted[(med-1)>>1] for odd spans
      for (int e = tid; e < m + med - 1; e += 256) {
        const float c = s_win[e];
        const int fi0 = max(0, e - (med - 1)), fi1 = min(m - 1, e);
        int lt = 0, le = 0;
        for (int k = 0; k < med; ++k) {
          const float w = s_win[fi0 + k];
          lt += (w < c) ? 1 : 0;
          le += (w <= c) ? 1 : 0;
        }
        for (int fi = fi0; fi <= fi1; ++fi) {
          if (lt <= r_lo && r_lo < le) s_vlo[fi] = c;
          if (lt <= r_hi && r_hi < le) s_vhi[fi] = c;
          if (fi < fi1) {
            const float gone = s_win[fi], come = s_win[fi + med];
            lt += ((come < c) ? 1 : 0) - ((gone < c) ? 1 : 0);
            le += ((come <= c) ? 1 : 0) - ((gone <= c) ? 1 : 0);
          }
        }
      }
      __syncthreads();
      float post = 0.f;
      if (tid < m) {
        const float v_lo = s_vlo[tid], v_hi = s_vhi[tid];
        const float median = (med & 1) ? v_hi : ((v_hi + v_lo) * 0.5f);
        post = s_win[tid + med - 1] - median;      // this frame's value is the newest of its span
        s_post[tid] = post;
      }
      const float carried = s_post[256];        // mOdfvalpostprev of the chunk's first frame
      __syncthreads();
      {
        // candidates (post > thresh, previous <= thresh) as one 64-bit mask per wave; the gap rule (mGapLeft) walks the
        // set bits only: after a detection at frame p the next one is allowed from frame p + mingap + 1
        const float prev = (tid == 0) ? carried : s_post[(tid > 0) ? tid - 1 : 0];
        const bool cand = (tid < m) && (post > thresh) && (prev <= thresh);
        const unsigned long long mask = __ballot(cand);
        if ((tid & 63) == 0) s_mask[tid >> 6] = mask;
      }
      __syncthreads();
      if (tid == 0) {
        int next_allowed = s_int[0];
        for (int w = 0; w < 4; ++w) {
          unsigned long long mk = s_mask[w], det = 0;
          while (mk) {
            const int bit = __ffsll((long long)mk) - 1;
            mk &= mk - 1;
            const int fr = c0 + 64 * w + bit;
            if (fr >= next_allowed) { det |= 1ull << bit; next_allowed = fr + mingap + 1; }
          }
          s_mask[w] = det;
        }
        s_int[0] = next_allowed;
        s_post[256] = s_post[m - 1];
      }
      __syncthreads();
      if (tid < m) {
        const double v = ((s_mask[tid >> 6] >> (tid & 63)) & 1) ? (double)post : 0.0;     // RT.cpp:109-118
        raw[c0 + tid] = v;
        a.onsets[(f.frame0 + c0 + tid) * 2 + type] = v;
        if (v > (type == 0 ? 0.2 : 0.8)) atomicAdd(&s_int[1], 1);   // OnsetCount, RT.cpp:124-137
      }
      __syncthreads();
    }
    const int count = s_int[1];
    __syncthreads();

    // ---- TCannyWindow::Apply (CW.cpp:27-68) ----
    for (int i = tid; i < T; i += 256) {
      double sum = 0.0;
      for (int s = -12; s < 12; ++s)
        if (i + s >= 0 && i + s < T) sum += raw[i + s] * a.canny[s + 12];
      W[i] = sum;
    }
    __syncthreads();
    const double w_first = W[0];
    const double mean = mean_of(serial_sum(T, [&](int i) { return W[i]; }, s_stage, s_result), T, w_first);
    const double var = (T >= 2) ? serial_sum(T, [&](int i) { return (W[i] - mean) * (W[i] - mean); }, s_stage, s_result) / T : 0.0;
    {
      const double sd = sqrt(var);
      for (int i = tid; i < T; i += 256) {
        double z = W[i];
        if (var > 0.0) {
          z = (z - mean) / sd;
          z = (0.0 > z) ? 0.0 : z;
        }
        S[i] = z;
      }
    }
    if (tid == 0) s_int[2] = 0;
    __syncthreads();

    // ---- CalculateTempo (RT.cpp:159-234): one aubio_beattracking_do on a fresh tracker (bt.c:132-186, 273-404) ----
    if (count >= 4) {
      const unsigned laglen = (unsigned)T / 4;
      const int plane = (a.lds_frames + 3) >> 2;
      if (T <= a.lds_frames) {
        for (int i = tid; i < T; i += 256) s_series[(i & 3) * plane + (i >> 2)] = S[i];
        __syncthreads();
        autocorrelation<true>(S, s_series, plane, T, W);
      } else autocorrelation<false>(S, s_series, plane, T, W);
      __syncthreads();
      const double ray = 60. * rate / 120. / kRtHop;           // bt.c:65
      for (unsigned i = tid; i < laglen; i += 256) {
        double acc = 0.0;
        if (i >= 1 && i + 1 < laglen)                          // comb filterbank, bt.c:165-172
          for (unsigned aa = 1; aa <= 4; ++aa)
            for (unsigned b = 1; b < 2 * aa; ++b) acc += W[i * aa + b - 1] * 1. / (2. * aa - 1.);
        const double rw = ((int)i < a.rayleigh_n)                // Rayleigh weight, bt.c:105-108, 174
                              ? a.rayleigh[i]
                              : ((double)(i + 1.) / (ray * ray)) * exp((-((i + 1.) * (i + 1.)) / (2. * (ray * ray))));
        AO[i] = acc * rw;
      }
      __syncthreads();
      // fvec_max_elem (mathutils.c:268-283) starts from 0 and lets later equals win: the last index holding max(0, max)
      {
        double best = 0.0;
        int pos = 0;
        for (unsigned i = tid; i < laglen; i += 256) {
          const double d = AO[i];
          if (!(best > d)) { best = d; pos = (int)i; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {       // larger value wins, the later index among equals
          const double d = __shfl_xor(best, o);
          const int p = __shfl_xor(pos, o);
          if (d > best || (d == best && p > pos)) { best = d; pos = p; }
        }
        if ((tid & 63) == 0) { s_stage[tid >> 6] = best; s_hist[tid >> 6] = (unsigned)pos; }
        __syncthreads();
        if (tid == 0) {
          double bv = 0.0;
          unsigned bp = 0;
          for (int t = 0; t < 4; ++t) {
            const double d = s_stage[t];
            const unsigned p = s_hist[t];
            if (d > bv || (d == bv && p > bp)) { bv = d; bp = p; }
          }
          s_int[3] = (int)bp;
        }
        __syncthreads();
      }
      const unsigned maxindex = (unsigned)s_int[3];
      __syncthreads();
      const double acf_sum = serial_sum((int)laglen, [&](int i) { return AO[i]; }, s_stage, s_result);
      if (tid == 0) {
        double rp;
        if (maxindex > 0 && laglen > 0 && maxindex < laglen - 1) {   // fvec_quadratic_peak_pos, mathutils.c:494-506
          const double s0 = AO[maxindex - 1], s1 = AO[maxindex], s2 = AO[maxindex + 1];
          rp = maxindex + .5 * (s0 - s2) / (s0 - 2. * s1 + s2);
        } else rp = (double)(unsigned)ray;                     // p->rayparam is uint_t, bt.c:46, 82
        double bp = rp;                                        // checkstate on the fresh state, bt.c:362-379
        while (0 < bp && bp < 25) bp = bp * 2;
        double bpm = (bp != 0) ? 60. / (kRtHop * bp / (double)rate) : 0.;
        double cf = 0.;                                        // bt.c:435-444
        if (acf_sum != 0.) {                                   // fvec_quadratic_peak_mag, mathutils.c:508-517
          double mag = 0.;
          if (!(rp >= laglen || rp < 0.)) {
            const unsigned index = (unsigned)(rp - .5) + 1;
            if ((double)index == rp) mag = AO[index];
            else mag = AO[index] - .25 * (AO[index - 1] - AO[index + 1]) * (rp - index);
          }
          cf = mag / acf_sum;
        }
        cf = cf * 16.0;
        cf = (cf < 0.0) ? 0.0 : ((cf > 1.0) ? 1.0 : cf);
        if (bpm < 20.0 || bpm > 300.0) { cf = 0.0; bpm = 0.0; }
        else {
          while (bpm < 80.0) bpm *= 2.0;
          while (bpm >= 200.0) bpm /= 2.0;
        }
        s_result[0] = bpm;
        s_result[1] = cf;
      }
      __syncthreads();
      tempo[type] = s_result[0];
      conf[type] = s_result[1];
      __syncthreads();
    }

    // ---- peaks of the sharpened onsets (RT.cpp:624-660): W <- 1 where no larger value lies within +-24 frames ----
    for (int i = tid; i < T; i += 256) {
      W[i] = window_peak(S, T, i) ? 1.0 : 0.0;
      if (i >= 1 && !(S[i] < 0.1)) atomicMax(&s_int[2], i);   // last frame that is not below MPeakThreshold, RT.cpp:264-268
    }
    if (tid == 0) s_int[1] = 0;
    __syncthreads();
    last_loud[type] = s_int[2];
    for (int i = tid; i < T; i += 256)
      if (S[i] > 0.1 && W[i] != 0.0) atomicAdd(&s_int[1], 1);
    __syncthreads();
    const int n_peaks = s_int[1];
    const double peak_sum = serial_sum(T, [&](int i) { return (S[i] > 0.1 && W[i] != 0.0) ? S[i] : 0.0; }, s_stage, s_result);
    const double total_sum = serial_sum(T, [&](int i) { return S[i]; }, s_stage, s_result);

    // ---- CalculateRhythmContrast (RT.cpp:325-412) ----
    // threshold = sorted[(int)(0.85 (T - 1))]: radix select on the order-preserving bit pattern, 8 bits per pass
    {
      int k = (int)(85.0 / 100.0 * (T - 1));
      unsigned long long prefix = 0;
      for (int pass = 0; pass < 8; ++pass) {
        const int shift = 56 - 8 * pass;
        s_hist[tid] = 0;
        __syncthreads();
        for (int i = tid; i < T; i += 256) {
          const unsigned long long key = order_key(S[i]);
          if (pass == 0 || (key >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&s_hist[(key >> shift) & 255], 1u);
        }
        __syncthreads();
        {
          // the digit whose bin holds the k-th element: exclusive prefix sum of the 256 bins, one bin per thread
          const int h = (int)s_hist[tid];
          const int incl = wave_scan_incl(h);
          if ((tid & 63) == 63) s_wsum[tid >> 6] = incl;
          __syncthreads();
          int before = 0;
          for (int w = 0; w < (tid >> 6); ++w) before += s_wsum[w];
          const int excl = before + incl - h;
          if (h > 0 && excl <= k && k < excl + h) {
            s_int[3] = k - excl;
            s_key = prefix | ((unsigned long long)tid << shift);
          }
        }
        __syncthreads();
        k = s_int[3];
        prefix = s_key;
        __syncthreads();
      }
      const double threshold = key_value(prefix);
      // serial walk with the running valley (RT.cpp:343-383)
      if (tid == 0) { s_result[0] = 0.0; s_result[1] = 0.0; s_int[3] = 0; }
      double valley_value = threshold, valley_at = S[0], psum = 0.0, vsum = 0.0;   // thread 0's copies are the live ones
      int np = 0;
      for (int c0 = 0; c0 < T; c0 += kStage) {
        const int m = min(kStage, T - c0);
        __syncthreads();
        // stage the values and, one bit per frame, "at or above the threshold and a window peak"
        for (int i = tid; i < kStage; i += 256) {
          const bool in = i < m;
          const double sv = in ? S[c0 + i] : 0.0;
          s_stage[i] = sv;
          const bool flag = in && !(sv < threshold) && W[c0 + i] != 0.0;
          const unsigned long long bits = __ballot(flag);
          if ((tid & 63) == 0) { s_flagbits[i >> 5] = (unsigned)bits; s_flagbits[(i >> 5) + 1] = (unsigned)(bits >> 32); }
        }
        __syncthreads();
        if (tid == 0) {
          for (int i0 = 0; i0 < m; i0 += 8) {
            const double2 p0 = *reinterpret_cast<const double2*>(s_stage + i0), p1 = *reinterpret_cast<const double2*>(s_stage + i0 + 2),
                          p2 = *reinterpret_cast<const double2*>(s_stage + i0 + 4), p3 = *reinterpret_cast<const double2*>(s_stage + i0 + 6);
            const double xs[8] = {p0.x, p0.y, p1.x, p1.y, p2.x, p2.y, p3.x, p3.y};
            const unsigned fl = (s_flagbits[i0 >> 5] >> (i0 & 31)) & 0xFFu;
            const int cnt = min(8, m - i0);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              if (k < cnt) {
                const double xv = xs[k];
                if (xv < valley_value) { valley_at = xv; valley_value = xv; }
                if ((fl >> k) & 1) {
                  psum += xv;
                  vsum += valley_at;
                  ++np;
                  valley_value = xv;
                }
              }
            }
          }
        }
      }
      if (tid == 0) {
        const double total_mean = mean_of(total_sum, T, S[0]);
        // Mean of a single-element list is the element; psum / vsum hold exactly that element then
        const double peak_mean = mean_of(psum, np, psum);
        const double valley_mean = mean_of(vsum, np, vsum) + 0.0001;
        double contrast = 0.0;
        if (peak_mean != 0.0) contrast = -1.0 * pow(peak_mean / valley_mean, 1.0 / log(total_mean + 0.0001));
        o6[0] = (double)count;
        o6[1] = tempo[type];
        o6[2] = conf[type];
        o6[3] = (double)n_peaks / (double)T * (double)kRtHop / (double)512;      // RT.cpp:288-289
        double strength = 0.0;                                                  // RT.cpp:316-318
        if (n_peaks) {
          strength = mean_of(peak_sum, n_peaks, peak_sum) / 4.0;
          strength = (strength < 0.0) ? 0.0 : ((strength > 1.0) ? 1.0 : strength);
        }
        o6[4] = strength;
        o6[5] = contrast;
      }
      __syncthreads();
    }
  }

  // the last analysed frame above the threshold travels to rhythm_final_kernel in the slot of the final results
  if (tid == 0) out[12 + (blockIdx.x & 1)] = (double)last_loud[blockIdx.x & 1];
}

// ---- final tempo: CalculateTempoWithHeuristics on the more confident function (SampleAnalyser.cpp:1029-1048, RT.cpp:238-325);
//      one thread per file, after both chains of the file ----
__global__ __launch_bounds__(256) void rhythm_final_kernel(RhythmArgs a) {
  const int file = blockIdx.x * blockDim.x + threadIdx.x;
  if (file >= a.n_files) return;
  const RhythmFile f = a.files[file];
  const int T = f.frames;
  double* out = a.scalars + (int64_t)file * 14;
  if (T <= 0) { out[12] = 0.0; out[13] = 0.0; return; }
  const int64_t tot = a.total_frames;
  const int rate = a.sample_rate;
  const double tempo[2] = {out[1], out[7]}, conf[2] = {out[2], out[8]};
  const int last_loud[2] = {(int)out[12], (int)out[13]};
  {
    const int type = (conf[1] > conf[0]) ? 1 : 0;
    const double* raw = a.scratch + (int64_t)(0 + type) * tot + f.frame0;
    double t_out = 0.0, c_out = 0.0;
    if (tempo[type] != 0) {
      double tp = tempo[type];
      c_out = conf[type];
      const double samples_per_beat = 60.0 / tp * (unsigned)rate;
      const unsigned n_samples = (unsigned)last_loud[type] * (unsigned)kRtHop;
      if ((double)n_samples < samples_per_beat * 3) {
        c_out = 0.0;
        tp = 0.0;
      } else {
        // GuessNumberOfBeatsFromDuration(80, 180, duration), RT.cpp:434-486
        const double duration = f.duration_s, max_beat = 60.0 / 80.0, max_bar = 4.0 * max_beat;
        double beats = 0.0;
        if (duration < max_beat) beats = 0.0;
        else if (duration < max_bar) {
          beats = 4.0;
          for (int divider = 2; divider >= 1; divider /= 2) {
            if (duration < ((double)4 / (double)divider) * max_beat) { beats = 4.0 / (double)divider; break; }
          }
        } else {
          for (int bars = 1; bars <= 8; bars *= 2) {
            const double nb = 4.0 * bars;
            if (duration / nb < max_beat) { beats = nb; break; }
          }
        }
        if (beats >= 4 && beats <= 16) {
          const double guessed = beats / (duration / 60);
          const double delay = (double)((float)(kRtHop / 2) / ((float)rate / 1000.0f)) / 1000.0;   // SamplesToMs, AudioMath.inl:134-137
          // OnsetMatchConfidence, RT.cpp:490-540
          const float ms = (float)((f.offset_s + delay) * 1000);
          const float fv = (float)rate / 1000.0f * ms;                       // MsToSamples, AudioMath.inl:127-130
          const int offset_samples = (int)(fv + (signbit(fv) ? -0.5f : 0.5f));
          const double spb = 60.0 / guessed * (unsigned)rate;
          const int range = (int)((unsigned)(int)(spb / 32) / (unsigned)kRtHop);
          const double thr = (type == 0) ? 0.2 : 0.8;
          double strength = 0;
          for (int i = 0; i < beats * 2; ++i) {
            const int st = (int)(i * spb / 2.0) + offset_samples;
            const int idx = (st + kRtHop / 2) / kRtHop;
            double peak = 0.0;
            for (int j = idx - range; j < idx + range; ++j)
              if (j >= 0 && j < T) peak = (peak > raw[j]) ? peak : raw[j];
            if (peak >= thr) strength += 1.0;
          }
          const double r = strength / (beats * 2) * 2.0;
          const double g = (1.0 < r) ? 1.0 : r;
          if ((g > 0.5) || (c_out < 0.1 && g > 0.1) || (c_out < 0.5 && fabs(guessed - tp) < 10)) {
            tp = guessed;
            c_out = (0.5 > g) ? 0.5 : g;
          }
        }
      }
      t_out = tp;
    }
    out[12] = t_out;
    out[13] = c_out;
  }
}

}  // namespace

hipError_t launch_rhythm(const RhythmArgs& a, hipStream_t stream) {
  if (a.n_files <= 0) return hipSuccess;
  if (a.total_frames > 0) {
    if (a.n_long < a.n_files) {     // (every file long: nothing for the one-workgroup-per-file kernel)
      if (a.pcm_dtype == kPcmF64) hipLaunchKernelGGL((onset_function_kernel<double, false>), dim3(a.n_files), dim3(256), 0, stream, a);
      else if (a.pcm_dtype == kPcmScaledF32) hipLaunchKernelGGL((onset_function_kernel<float, true>), dim3(a.n_files), dim3(256), 0, stream, a);
      else hipLaunchKernelGGL((onset_function_kernel<float, false>), dim3(a.n_files), dim3(256), 0, stream, a);
    }
    if (a.n_long > 0) {
      if (a.pcm_dtype == kPcmF64) hipLaunchKernelGGL((onset_polar_kernel<double, false>), dim3(a.long_rounds), dim3(256), 0, stream, a);
      else if (a.pcm_dtype == kPcmScaledF32) hipLaunchKernelGGL((onset_polar_kernel<float, true>), dim3(a.long_rounds), dim3(256), 0, stream, a);
      else hipLaunchKernelGGL((onset_polar_kernel<float, false>), dim3(a.long_rounds), dim3(256), 0, stream, a);
      hipLaunchKernelGGL(onset_follow_kernel, dim3(a.n_long), dim3(256), 0, stream, a);
      hipLaunchKernelGGL(onset_terms_kernel, dim3(a.long_rounds), dim3(256), 0, stream, a);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  const size_t lds = (size_t)((a.lds_frames + 3) / 4) * 4 * sizeof(double);
  static bool attribute_set[16] = {};   // per device: raising the dynamic LDS limit once is enough
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 16 || !attribute_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rhythm_post_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)(kMaxLdsFrames * sizeof(double)));
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 16) attribute_set[dev] = true;
  }
  hipLaunchKernelGGL(rhythm_post_kernel, dim3(2 * a.n_files), dim3(256), lds, stream, a);
  hipLaunchKernelGGL(rhythm_final_kernel, dim3((a.n_files + 255) / 256), dim3(256), 0, stream, a);
  return hipGetLastError();
}

}  // namespace afx
