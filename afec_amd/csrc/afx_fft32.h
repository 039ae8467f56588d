// afec_amd/csrc/afx_fft32.h -- in-register 32-point DFT building blocks of the half-wave FFT
// (afx_frames32.hip): a 1024-point complex FFT as 32 x 32 with 32 complex values per lane and 32
// lanes per frame, so that a single LDS exchange separates the two passes.
//
// Per-lane arithmetic only (no cross-lane traffic, no memory): the header also compiles for the host
// (tests/host/test_fft32_math.cpp checks it against a direct DFT).
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define AFX_HD __device__ __forceinline__
#else
#define AFX_HD inline
#endif

namespace afx {
namespace f32x32 {

template <typename T>
struct cx {
  T re, im;
};

template <typename T>
AFX_HD cx<T> cmul(cx<T> a, cx<T> b) {
  return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}

// forward (e^{-i}) radix-4 butterfly, in place: (a,b,c,d) -> (X0,X1,X2,X3)
template <typename T>
AFX_HD void radix4(cx<T>& a, cx<T>& b, cx<T>& c, cx<T>& d) {
  const cx<T> t0{a.re + c.re, a.im + c.im}, t1{a.re - c.re, a.im - c.im};
  const cx<T> t2{b.re + d.re, b.im + d.im}, t3{b.re - d.re, b.im - d.im};
  a = {t0.re + t2.re, t0.im + t2.im};
  c = {t0.re - t2.re, t0.im - t2.im};
  b = {t1.re + t3.im, t1.im - t3.re};
  d = {t1.re - t3.im, t1.im + t3.re};
}

// x = e + w o, y = e - w o for w = (c, -s) = e^{-i theta}: four fused multiply-adds for x, then
// y = 2 e - x (two more) instead of a complex product (4) and two complex additions (4)
template <typename T>
AFX_HD void bfly_tw(cx<T>& e, cx<T>& o, T c, T s) {
  const T xr = fma(c, o.re, fma(s, o.im, e.re));
  const T xi = fma(c, o.im, fma(-s, o.re, e.im));
  o = {fma((T)2, e.re, -xr), fma((T)2, e.im, -xi)};
  e = {xr, xi};
}

// 16-point forward DFT in registers, natural order in and out
template <typename T>
AFX_HD void dft16(cx<T>& v0, cx<T>& v1, cx<T>& v2, cx<T>& v3, cx<T>& v4, cx<T>& v5, cx<T>& v6, cx<T>& v7,
                  cx<T>& v8, cx<T>& v9, cx<T>& v10, cx<T>& v11, cx<T>& v12, cx<T>& v13, cx<T>& v14, cx<T>& v15) {
  constexpr T c1 = T(0.92387953251128673848), s1 = T(0.38268343236508978178);
  constexpr T rh = T(0.70710678118654752440);
  radix4(v0, v4, v8, v12);
  radix4(v1, v5, v9, v13);
  radix4(v2, v6, v10, v14);
  radix4(v3, v7, v11, v15);
  // now v[4c + b] = y[b][c]; multiply by w16^(b c)
  v5 = cmul(v5, cx<T>{c1, -s1});
  v9 = {(v9.re + v9.im) * rh, (v9.im - v9.re) * rh};
  v13 = cmul(v13, cx<T>{s1, -c1});
  v6 = {(v6.re + v6.im) * rh, (v6.im - v6.re) * rh};
  v10 = {v10.im, -v10.re};
  v14 = {(v14.im - v14.re) * rh, -(v14.re + v14.im) * rh};
  v7 = cmul(v7, cx<T>{s1, -c1});
  v11 = {(v11.im - v11.re) * rh, -(v11.re + v11.im) * rh};
  v15 = cmul(v15, cx<T>{-c1, s1});
  radix4(v0, v1, v2, v3);
  radix4(v4, v5, v6, v7);
  radix4(v8, v9, v10, v11);
  radix4(v12, v13, v14, v15);
  // v[4c + d] = X[c + 4d]: transpose the 4x4 register grid (pure renaming)
  cx<T> t;
  t = v1; v1 = v4; v4 = t;
  t = v2; v2 = v8; v8 = t;
  t = v3; v3 = v12; v12 = t;
  t = v6; v6 = v9; v9 = t;
  t = v7; v7 = v13; v13 = t;
  t = v11; v11 = v14; v14 = t;
}

// 32-point forward DFT in registers, natural order in and out: two 16-point DFTs over the even and
// the odd inputs, then sixteen twiddled radix-2 butterflies (w32^k known at compile time).
template <typename T>
AFX_HD void dft32(cx<T> (&v)[32]) {
  dft16(v[0], v[2], v[4], v[6], v[8], v[10], v[12], v[14], v[16], v[18], v[20], v[22], v[24], v[26], v[28], v[30]);
  dft16(v[1], v[3], v[5], v[7], v[9], v[11], v[13], v[15], v[17], v[19], v[21], v[23], v[25], v[27], v[29], v[31]);
  // E[k] = v[2k], O[k] = v[2k+1]:  X[k] = E[k] + w32^k O[k] -> v[2k],  X[k+16] = E[k] - w32^k O[k] -> v[2k+1]
  constexpr T rh = T(0.70710678118654752440);
  constexpr T c[8] = {T(1.0), T(0.98078528040323044913), T(0.92387953251128675613), T(0.83146961230254523708),
                      rh, T(0.55557023301960222474), T(0.38268343236508977173), T(0.19509032201612826785)};
  constexpr T s[8] = {T(0.0), T(0.19509032201612826785), T(0.38268343236508977173), T(0.55557023301960222474),
                      rh, T(0.83146961230254523708), T(0.92387953251128675613), T(0.98078528040323044913)};
  {  // k = 0: w = 1
    const cx<T> e = v[0], o = v[1];
    v[0] = {e.re + o.re, e.im + o.im};
    v[1] = {e.re - o.re, e.im - o.im};
  }
  bfly_tw(v[2], v[3], c[1], s[1]);
  bfly_tw(v[4], v[5], c[2], s[2]);
  bfly_tw(v[6], v[7], c[3], s[3]);
  {  // k = 4: w = rh (1 - i)
    const cx<T> e = v[8], o = v[9];
    const T u = o.re + o.im, d = o.im - o.re;
    v[8] = {fma(rh, u, e.re), fma(rh, d, e.im)};
    v[9] = {fma(-rh, u, e.re), fma(-rh, d, e.im)};
  }
  bfly_tw(v[10], v[11], c[5], s[5]);
  bfly_tw(v[12], v[13], c[6], s[6]);
  bfly_tw(v[14], v[15], c[7], s[7]);
  {  // k = 8: w = -i
    const cx<T> e = v[16], o = v[17];
    v[16] = {e.re + o.im, e.im - o.re};
    v[17] = {e.re - o.im, e.im + o.re};
  }
  // k = 8 + j: w32^(8+j) = -i w32^j = (-s_j, -c_j) = (c', -s') with c' = -s_j, s' = c_j
  bfly_tw(v[18], v[19], -s[1], c[1]);
  bfly_tw(v[20], v[21], -s[2], c[2]);
  bfly_tw(v[22], v[23], -s[3], c[3]);
  {  // k = 12: w = -rh (1 + i)
    const cx<T> e = v[24], o = v[25];
    const T u = o.re + o.im, d = o.im - o.re;
    v[24] = {fma(rh, d, e.re), fma(-rh, u, e.im)};
    v[25] = {fma(-rh, d, e.re), fma(rh, u, e.im)};
  }
  bfly_tw(v[26], v[27], -s[5], c[5]);
  bfly_tw(v[28], v[29], -s[6], c[6]);
  bfly_tw(v[30], v[31], -s[7], c[7]);
  // v[2k] = X[k], v[2k+1] = X[k+16]: un-interleave (pure renaming)
  cx<T> t[32];
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int k = 0; k < 16; ++k) {
    t[k] = v[2 * k];
    t[k + 16] = v[2 * k + 1];
  }
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int i = 0; i < 32; ++i) v[i] = t[i];
}

}  // namespace f32x32
}  // namespace afx
