// afec_amd/csrc/afx_device.h -- wave-level device helpers shared by the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>

namespace afx {
namespace {

// orders this wave's LDS traffic for the compiler; the DS unit executes a wave's ops in order
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// cooperative copy of a constant table into LDS (bytes a multiple of 16)
__device__ __forceinline__ void copy_lds_table(unsigned char* dst, const void* src, int bytes, int tid, int nthreads) {
  const uint4* s = reinterpret_cast<const uint4*>(src);
  uint4* d = reinterpret_cast<uint4*>(dst);
  for (int i = tid; i < bytes / 16; i += nthreads) d[i] = s[i];
}

// ---- cross-lane helpers on doubles: DPP inside a 16-lane row, permlane swaps across rows ----
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
constexpr int kDppXor1 = 0xB1;        // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;        // quad_perm [2,3,0,1]
constexpr int kDppHalfMirror = 0x141; // i -> 7 - i inside groups of 8
constexpr int kDppMirror = 0x140;     // i -> 15 - i inside the row
constexpr int kDppRor8 = 0x128;       // i -> i ^ 8 inside the row

// x' = (x.lanes[0:32), y.lanes[0:32)), y' = (x.lanes[32:64), y.lanes[32:64))
__device__ __forceinline__ void swap32(double& x, double& y) {
  const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(x), __double2loint(y), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(x), __double2hiint(y), false, false);
  x = __hiloint2double(hi[0], lo[0]);
  y = __hiloint2double(hi[1], lo[1]);
}
// rows of 16: x' = (x0, y0, x2, y2), y' = (x1, y1, x3, y3)
__device__ __forceinline__ void swap16(double& x, double& y) {
  const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(x), __double2loint(y), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(x), __double2hiint(y), false, false);
  x = __hiloint2double(hi[0], lo[0]);
  y = __hiloint2double(hi[1], lo[1]);
}

__device__ __forceinline__ void swap32(float& x, float& y) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  x = __uint_as_float(r[0]);
  y = __uint_as_float(r[1]);
}
__device__ __forceinline__ void swap16(float& x, float& y) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  x = __uint_as_float(r[0]);
  y = __uint_as_float(r[1]);
}
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

// Sum 16 per-lane values over the wave at once ("transposed" butterfly): on return lane L holds
// the wave total of a[(L >> 2) & 15].  15 adds + 24 swaps + a few DPP moves instead of 16 x 6 steps.
template <typename A>
__device__ __forceinline__ A wave_sum16(A (&a)[16], int lane) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    swap32(a[i], a[i + 8]);
    a[i] += a[i + 8];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    swap16(a[i], a[i + 4]);
    a[i] += a[i + 4];
  }
  const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const A keep = b3 ? a[i + 2] : a[i];
    const A send = b3 ? a[i] : a[i + 2];
    a[i] = keep + dpp_mov<kDppRor8>(send);
  }
  const A keep = b2 ? a[1] : a[0];
  const A send = b2 ? a[0] : a[1];
  A z = keep + dpp_mov<kDppHalfMirror>(send);
  z += dpp_mov<kDppXor2>(z);
  z += dpp_mov<kDppXor1>(z);
  return z;
}

// value of lane L (compile-time) as a wave-uniform double: two v_readlane_b32, no LDS traffic
template <int L>
__device__ __forceinline__ double read_lane(double v) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), L);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), L);
  return __hiloint2double(hi, lo);
}
// full-wave sum, result in every lane: DPP inside the 16-lane rows, readlane across the four rows
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_mov<kDppXor1>(v);
  v += dpp_mov<kDppXor2>(v);
  v += dpp_mov<kDppHalfMirror>(v);
  v += dpp_mov<kDppMirror>(v);
  return (read_lane<0>(v) + read_lane<16>(v)) + (read_lane<32>(v) + read_lane<48>(v));
}
__device__ __forceinline__ double wave_max(double v) {
  v = fmax(v, dpp_mov<kDppXor1>(v));
  v = fmax(v, dpp_mov<kDppXor2>(v));
  v = fmax(v, dpp_mov<kDppHalfMirror>(v));
  v = fmax(v, dpp_mov<kDppMirror>(v));
  return fmax(fmax(read_lane<0>(v), read_lane<16>(v)), fmax(read_lane<32>(v), read_lane<48>(v)));
}
__device__ __forceinline__ double wave_min(double v) {
  v = fmin(v, dpp_mov<kDppXor1>(v));
  v = fmin(v, dpp_mov<kDppXor2>(v));
  v = fmin(v, dpp_mov<kDppHalfMirror>(v));
  v = fmin(v, dpp_mov<kDppMirror>(v));
  return fmin(fmin(read_lane<0>(v), read_lane<16>(v)), fmin(read_lane<32>(v), read_lane<48>(v)));
}
// Cross-lane sums, minima, maxima and prefix sums below move data with DPP and v_readlane only: a ds_bpermute step is an
// LDS round trip (~100 cycles), and six dependent ones per reduction were exposed latency at two waves per SIMD.
// v of the lane the control names, 0 where it names none (or the row is not in ROW_MASK)
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ int dpp_or_zero(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xF, false); }
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ double dpp_or_zero(double v) {
  return __hiloint2double(dpp_or_zero<CTRL, ROW_MASK>(__double2hiint(v)), dpp_or_zero<CTRL, ROW_MASK>(__double2loint(v)));
}
constexpr int kDppRowShr1 = 0x111, kDppRowShr2 = 0x112, kDppRowShr4 = 0x114, kDppRowShr8 = 0x118;
constexpr int kDppRowBcast15 = 0x142, kDppRowBcast31 = 0x143;   // lane 15 of a row to the next row; lane 31 to rows 2, 3

// v of lane + 1 (wave_shl:1: lane i reads lane i + 1); lane 63 gets `last`
__device__ __forceinline__ int next_lane(int v, int last) { return __builtin_amdgcn_update_dpp(last, v, 0x130, 0xF, 0xF, false); }
__device__ __forceinline__ float next_lane(float v, float last) { return __int_as_float(next_lane(__float_as_int(v), __float_as_int(last))); }
__device__ __forceinline__ double next_lane(double v, double last) {
  return __hiloint2double(next_lane(__double2hiint(v), __double2hiint(last)), next_lane(__double2loint(v), __double2loint(last)));
}
// v of lane - 1 (wave_shr:1); lane 0 gets `first`
__device__ __forceinline__ int prev_lane(int v, int first) { return __builtin_amdgcn_update_dpp(first, v, 0x138, 0xF, 0xF, false); }
__device__ __forceinline__ double prev_lane(double v, double first) {
  return __hiloint2double(prev_lane(__double2hiint(v), __double2hiint(first)), prev_lane(__double2loint(v), __double2loint(first)));
}

__device__ __forceinline__ int wave_sum_i(int v) {
  v += dpp_or_zero<kDppXor1>(v);
  v += dpp_or_zero<kDppXor2>(v);
  v += dpp_or_zero<kDppHalfMirror>(v);
  v += dpp_or_zero<kDppMirror>(v);
  return (__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)) + (__builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
}
__device__ __forceinline__ int wave_min_i(int v) {
  v = min(v, __builtin_amdgcn_update_dpp(v, v, kDppXor1, 0xF, 0xF, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, kDppXor2, 0xF, 0xF, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, kDppHalfMirror, 0xF, 0xF, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, kDppMirror, 0xF, 0xF, false));
  return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int wave_max_i(int v) {
  v = max(v, __builtin_amdgcn_update_dpp(v, v, kDppXor1, 0xF, 0xF, false));
  v = max(v, __builtin_amdgcn_update_dpp(v, v, kDppXor2, 0xF, 0xF, false));
  v = max(v, __builtin_amdgcn_update_dpp(v, v, kDppHalfMirror, 0xF, 0xF, false));
  v = max(v, __builtin_amdgcn_update_dpp(v, v, kDppMirror, 0xF, 0xF, false));
  return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
// inclusive prefix maximum across the 64 lanes (a lane without a source keeps its own value)
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ int dpp_or_self(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xF, false); }
__device__ __forceinline__ int wave_scan_max_i(int v) {
  v = max(v, dpp_or_self<kDppRowShr1>(v));
  v = max(v, dpp_or_self<kDppRowShr2>(v));
  v = max(v, dpp_or_self<kDppRowShr4>(v));
  v = max(v, dpp_or_self<kDppRowShr8>(v));
  v = max(v, dpp_or_self<kDppRowBcast15, 0xA>(v));
  v = max(v, dpp_or_self<kDppRowBcast31, 0xC>(v));
  return v;
}
// inclusive prefix sum across the 64 lanes: inside the rows of 16 (row_shr 1, 2, 4, 8), then row 0's total into row 1 and
// row 2's into row 3 (row_bcast:15), then the total of rows 0..1 into rows 2 and 3 (row_bcast:31)
template <typename T>
__device__ __forceinline__ T wave_scan_incl(T v, int /*lane*/ = 0) {
  v += dpp_or_zero<kDppRowShr1>(v);
  v += dpp_or_zero<kDppRowShr2>(v);
  v += dpp_or_zero<kDppRowShr4>(v);
  v += dpp_or_zero<kDppRowShr8>(v);
  v += dpp_or_zero<kDppRowBcast15, 0xA>(v);
  v += dpp_or_zero<kDppRowBcast31, 0xC>(v);
  return v;
}
// the same inside each half of the wave (lanes 0..31 and 32..63 on their own)
template <typename T>
__device__ __forceinline__ T halfwave_scan_incl(T v) {
  v += dpp_or_zero<kDppRowShr1>(v);
  v += dpp_or_zero<kDppRowShr2>(v);
  v += dpp_or_zero<kDppRowShr4>(v);
  v += dpp_or_zero<kDppRowShr8>(v);
  v += dpp_or_zero<kDppRowBcast15, 0xA>(v);
  return v;
}

__device__ __forceinline__ double nan_to_zero(double v) { return (v != v) ? 0.0 : v; }

// a double that is the same in every lane, moved to an SGPR pair (the compiler cannot prove it for values loaded through
// a per-wave index)
__device__ __forceinline__ double wave_uniform(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// the wave's next item of a WorkQueue launch (afx_internal.h): n_static = min(waves of the grid, items)
__device__ __forceinline__ int next_item(const WorkQueue& q, int current, int n_static, int stride, int lane) {
  if (!q.counter) return current + stride;
  unsigned drawn = 0;
  if (lane == 0) drawn = atomicAdd(q.counter, 1u);
  return n_static + (int)((unsigned)__builtin_amdgcn_readfirstlane((int)drawn) - q.base);
}

// a * b rounded to a double of its own: never fused into the addition that consumes it.  For values that two code paths
// must compute alike (a sum carried from frame to frame and the same sum recomputed where a chunk starts): whether the
// compiler contracts a product into an fma depends on the code around it.
__device__ __forceinline__ double mul_rn(double a, double b) {
#pragma clang fp contract(off)
  return a * b;
}

// A PCM sample as the double the reference's loop sees.  SCALED (kPcmScaledF32, afx_internal.h): the arena holds the
// float mono signal LoadSample worked on and `scale` is the buffer's FinalScaling -- the product, rounded once, is the
// reference's TSampleData::mData[n] bit for bit (SampleAnalyser.cpp:710-718).
// No contraction: fused into a later addition (fma(x, scale, b)) the product would be rounded once less than the
// reference's stored double.
template <bool SCALED, typename X>
__device__ __forceinline__ double pcm_double(X x, double scale) {
#pragma clang fp contract(off)
  return SCALED ? (double)x * scale : (double)x;
}

// sqrt for magnitudes: x >= 0 and far from overflow, so the range scaling of the generic
// expansion is dropped: one v_rsq_f64 seed, one Goldschmidt step, one residual correction
// (<= 1 ulp).  Sums of squares below the smallest normal double are flushed to 0: the reference's
// TAudioMath::Magnitude runs with the SSE DAZ + FZ bits set (AudioMath.cpp:25-35, 478-480), so its
// bins below ~1.5e-154 are exactly 0 as well.
__device__ __forceinline__ double mag_sqrt(double x) {
  constexpr double kMinNormal = 2.2250738585072014e-308;
  const double xs = x >= kMinNormal ? x : kMinNormal;
  const double r = __builtin_amdgcn_rsq(xs);
  double g = xs * r;
  double h = 0.5 * r;
  const double e = fma(-h, g, 0.5);
  g = fma(g, e, g);
  h = fma(h, e, h);
  const double d = fma(-g, g, xs);
  g = fma(d, h, g);
  return x >= kMinNormal ? g : 0.0;
}
__device__ __forceinline__ float mag_sqrt(float x) { return sqrtf(x); }

// n / d for normal d of either sign and moderate n: v_rcp_f64 seed, two Newton steps, then one residual
// correction of the quotient (exact whenever n / d is representable, <= 1 ulp otherwise).  The generic division
// with its range scaling is ~25 instructions.
__device__ __forceinline__ double fast_div(double n, double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = fma(r, fma(-d, r, 1.0), r);
  r = fma(r, fma(-d, r, 1.0), r);
  const double q = n * r;
  return fma(fma(-d, q, n), r, q);
}


// transposed butterfly maximum, same lane map as wave_sum16: lane L gets max over the wave of a[(L >> 2) & 15]
__device__ __forceinline__ double wave_max16(double (&a)[16], int lane) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    swap32(a[i], a[i + 8]);
    a[i] = fmax(a[i], a[i + 8]);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    swap16(a[i], a[i + 4]);
    a[i] = fmax(a[i], a[i + 4]);
  }
  const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const double keep = b3 ? a[i + 2] : a[i];
    const double send = b3 ? a[i] : a[i + 2];
    a[i] = fmax(keep, dpp_mov<kDppRor8>(send));
  }
  const double keep = b2 ? a[1] : a[0];
  const double send = b2 ? a[0] : a[1];
  double z = fmax(keep, dpp_mov<kDppHalfMirror>(send));
  z = fmax(z, dpp_mov<kDppXor2>(z));
  z = fmax(z, dpp_mov<kDppXor1>(z));
  return z;
}

// natural log for positive normal doubles: frexp + atanh series (|s| <= 0.1716), ~1 ulp-level
// absolute accuracy on log(m); replaces the generic libm expansion in the per-element paths
__device__ __forceinline__ double fast_log(double x) {
  double m = __builtin_amdgcn_frexp_mant(x);   // [0.5, 1)
  int e = __builtin_amdgcn_frexp_exp(x);
  const bool lowhalf = m < 0.70710678118654752440;
  m = lowhalf ? m + m : m;
  e = lowhalf ? e - 1 : e;
  const double s = (m - 1.0) / (m + 1.0);
  const double z = s * s;
  double p = 1.0 / 23.0;
  p = fma(p, z, 1.0 / 21.0);
  p = fma(p, z, 1.0 / 19.0);
  p = fma(p, z, 1.0 / 17.0);
  p = fma(p, z, 1.0 / 15.0);
  p = fma(p, z, 1.0 / 13.0);
  p = fma(p, z, 1.0 / 11.0);
  p = fma(p, z, 1.0 / 9.0);
  p = fma(p, z, 1.0 / 7.0);
  p = fma(p, z, 1.0 / 5.0);
  p = fma(p, z, 1.0 / 3.0);
  p = fma(p, z, 1.0);
  return fma((double)e, 0.693147180559945309417, 2.0 * s * p);
}

// TAudioMath::LinToDb(double), AudioMath.inl:55-70 (MEpsilon is the float literal 1e-12f)
__device__ __forceinline__ double lin_to_db(double v) {
  if (v == 1.0) return 0.0;
  if (v > (double)1e-12f) return fast_log(v) * 8.685889638065035;  // 20 / ln 10
  return -200.0;
}

// e^x for |x| < 700: x = k ln2 + r, |r| <= ln2/2, degree-13 Taylor on r (|error| < 2e-16 relative), ldexp
__device__ __forceinline__ double fast_exp(double x) {
  const double kf = rint(x * 1.44269504088896340736);
  const double r = fma(-kf, 1.90821492927058770002e-10, fma(-kf, 6.93147180369123816490e-01, x));
  double p = 1.0 / 6227020800.0;
  p = fma(p, r, 1.0 / 479001600.0);
  p = fma(p, r, 1.0 / 39916800.0);
  p = fma(p, r, 1.0 / 3628800.0);
  p = fma(p, r, 1.0 / 362880.0);
  p = fma(p, r, 1.0 / 40320.0);
  p = fma(p, r, 1.0 / 5040.0);
  p = fma(p, r, 1.0 / 720.0);
  p = fma(p, r, 1.0 / 120.0);
  p = fma(p, r, 1.0 / 24.0);
  p = fma(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)kf);
}

}  // namespace
}  // namespace afx
