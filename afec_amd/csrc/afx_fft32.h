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

// radix-4 butterfly of (a, w1 b, w2 c, w3 d) with the three products fused into the additions:
//   t0 = a + w2 c (4 fma), t1 = 2 a - t0 (2), b1 = w1 b (4), t2 = b1 + w3 d (4), t3 = 2 b1 - t2 (2), outputs (8)
// = 24 operations instead of 3 complex products (12) + 16 additions.  w = (re, im) multiplies as a complex number.
template <typename T>
AFX_HD void radix4_tw3(cx<T>& a, cx<T>& b, cx<T>& c, cx<T>& d, cx<T> w1, cx<T> w2, cx<T> w3) {
  const T t0r = fma(w2.re, c.re, fma(-w2.im, c.im, a.re));
  const T t0i = fma(w2.re, c.im, fma(w2.im, c.re, a.im));
  const T t1r = fma((T)2, a.re, -t0r), t1i = fma((T)2, a.im, -t0i);
  const cx<T> b1 = cmul(b, w1);
  const T t2r = fma(w3.re, d.re, fma(-w3.im, d.im, b1.re));
  const T t2i = fma(w3.re, d.im, fma(w3.im, d.re, b1.im));
  const T t3r = fma((T)2, b1.re, -t2r), t3i = fma((T)2, b1.im, -t2i);
  a = {t0r + t2r, t0i + t2i};
  c = {t0r - t2r, t0i - t2i};
  b = {t1r + t3i, t1i - t3r};
  d = {t1r - t3i, t1i + t3r};
}
// the same with a fourth factor on a: 28 operations instead of 16 + 16
template <typename T>
AFX_HD void radix4_tw4(cx<T>& a, cx<T>& b, cx<T>& c, cx<T>& d, cx<T> w0, cx<T> w1, cx<T> w2, cx<T> w3) {
  a = cmul(a, w0);
  radix4_tw3(a, b, c, d, w1, w2, w3);
}

// the part of the 16-point DFT behind its first radix-4 stage: on entry v[4c + b] = y[b][c] (output c of the
// first-stage butterfly b); factors w16^(b c) fused into the second-stage butterflies; natural order out
template <typename T>
AFX_HD void dft16_rest(cx<T>& v0, cx<T>& v1, cx<T>& v2, cx<T>& v3, cx<T>& v4, cx<T>& v5, cx<T>& v6, cx<T>& v7,
                       cx<T>& v8, cx<T>& v9, cx<T>& v10, cx<T>& v11, cx<T>& v12, cx<T>& v13, cx<T>& v14, cx<T>& v15) {
  constexpr T c1 = T(0.92387953251128673848), s1 = T(0.38268343236508978178);
  constexpr T rh = T(0.70710678118654752440);
  radix4(v0, v1, v2, v3);                                                                  // c = 0: no factors
  radix4_tw3(v4, v5, v6, v7, cx<T>{c1, -s1}, cx<T>{rh, -rh}, cx<T>{s1, -c1});              // c = 1: w, w^2, w^3
  {                                                                                        // c = 2: w^2, w^4 = -i, w^6
    const cx<T> a = v8, B = v9, C = v10, D = v11;
    const T t0r = a.re + C.im, t0i = a.im - C.re;       // a + (-i) C
    const T t1r = a.re - C.im, t1i = a.im + C.re;
    const T b2r = B.re + B.im, b2i = B.im - B.re;       // w^2 B = rh (b2r, b2i)
    const T d2r = D.im - D.re, d2i = -(D.re + D.im);    // w^6 D = rh (d2r, d2i)
    const T s2r = b2r + d2r, s2i = b2i + d2i, s3r = b2r - d2r, s3i = b2i - d2i;
    v8 = {fma(rh, s2r, t0r), fma(rh, s2i, t0i)};
    v10 = {fma(-rh, s2r, t0r), fma(-rh, s2i, t0i)};
    v9 = {fma(rh, s3i, t1r), fma(-rh, s3r, t1i)};
    v11 = {fma(-rh, s3i, t1r), fma(rh, s3r, t1i)};
  }
  radix4_tw3(v12, v13, v14, v15, cx<T>{s1, -c1}, cx<T>{-rh, -rh}, cx<T>{-c1, s1});         // c = 3: w^3, w^6, w^9
  // v[4c + d] = X[c + 4d]: transpose the 4x4 register grid (pure renaming)
  cx<T> t;
  t = v1; v1 = v4; v4 = t;
  t = v2; v2 = v8; v8 = t;
  t = v3; v3 = v12; v12 = t;
  t = v6; v6 = v9; v9 = t;
  t = v7; v7 = v13; v13 = t;
  t = v11; v11 = v14; v14 = t;
}

// the sixteen twiddled radix-2 butterflies that merge the DFTs of the even and the odd inputs; on entry
// v[2k] = E[k], v[2k+1] = O[k]; natural order out
template <typename T>
AFX_HD void dft32_merge(cx<T> (&v)[32]) {
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

// everything of the 32-point DFT behind its first radix-4 stage (the butterflies on v[j], v[j+8], v[j+16],
// v[j+24], j = 0..7, which dft32 and the twiddled variant of the caller do themselves)
template <typename T>
AFX_HD void dft32_rest(cx<T> (&v)[32]) {
  dft16_rest(v[0], v[2], v[4], v[6], v[8], v[10], v[12], v[14], v[16], v[18], v[20], v[22], v[24], v[26], v[28], v[30]);
  dft16_rest(v[1], v[3], v[5], v[7], v[9], v[11], v[13], v[15], v[17], v[19], v[21], v[23], v[25], v[27], v[29], v[31]);
  dft32_merge(v);
}

// 32-point forward DFT in registers, natural order in and out: two 16-point DFTs over the even and the odd
// inputs (first radix-4 stage: j, j+8, j+16, j+24), then sixteen twiddled radix-2 butterflies
template <typename T>
AFX_HD void dft32(cx<T> (&v)[32]) {
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int j = 0; j < 8; ++j) radix4(v[j], v[j + 8], v[j + 16], v[j + 24]);
  dft32_rest(v);
}

}  // namespace f32x32
}  // namespace afx
