// afec_amd/csrc/afx_fft.h -- the in-register 1024-point complex FFT shared by the gfx950 kernels.
//
// 16 complex values per lane, element index = lane + 64 * register on the way in and on the way out:
//
//   P1  16-point DFT over the register index            P2  4-point DFT          P3  16-point DFT
//   E1  register transpose (v_permlane32/16_swap)       E2  exchange through the wave's LDS plane
//   T1  twiddles w64 (global table, one line per row)   T2  twiddles w1024 (LDS table)
//
// The index algebra and the conflict-free LDS map are modelled in tools/fft_dataflow_model.py; the
// stage list with the data flow is at the top of afx_kernels.hip.
//
// The per-lane arithmetic (cmul, radix4, dft16) also compiles for the host: tests/host/test_fft32_math.cpp checks it
// against a direct DFT in long double; everything that moves data between lanes is device code only.
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>

#include "afx_device.h"
#define AFX_FFT_HD __device__ __forceinline__
#define AFX_FFT_UNROLL _Pragma("unroll")
#else
#define AFX_FFT_HD inline
#define AFX_FFT_UNROLL
#endif

namespace afx {
namespace {

// E2 plane: 8-byte slots, slot = k1 + 65 n2 (k1 < 64, n2 < 16): write = lane part (4 jh + 65 n2)
// + immediate (16 j2 + jl), read = lane + immediate 65 n2.  complex<float> is one slot;
// complex<double> goes through the same plane twice (real parts, then imaginary parts).
constexpr int kPlaneSlots = 1040;

template <typename T>
struct cx {
  T re, im;
};

template <typename T>
AFX_FFT_HD cx<T> cmul(cx<T> a, cx<T> b) {
  return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}

// forward (e^{-i}) radix-4 butterfly, in place: (a,b,c,d) -> (X0,X1,X2,X3)
template <typename T>
AFX_FFT_HD void radix4(cx<T>& a, cx<T>& b, cx<T>& c, cx<T>& d) {
  const cx<T> t0{a.re + c.re, a.im + c.im}, t1{a.re - c.re, a.im - c.im};
  const cx<T> t2{b.re + d.re, b.im + d.im}, t3{b.re - d.re, b.im - d.im};
  a = {t0.re + t2.re, t0.im + t2.im};
  c = {t0.re - t2.re, t0.im - t2.im};
  b = {t1.re + t3.im, t1.im - t3.re};
  d = {t1.re - t3.im, t1.im + t3.re};
}

// 16-point forward DFT in registers: v[n] -> v[k]
template <typename T>
AFX_FFT_HD void dft16(cx<T> (&v)[16]) {
  constexpr T c1 = T(0.92387953251128673848), s1 = T(0.38268343236508978178);
  constexpr T rh = T(0.70710678118654752440);
AFX_FFT_UNROLL
  for (int b = 0; b < 4; ++b) radix4(v[b], v[4 + b], v[8 + b], v[12 + b]);
  // now v[4c + b] = y[b][c]; multiply by w16^(b c)
  v[4 * 1 + 1] = cmul(v[5], cx<T>{c1, -s1});
  v[4 * 2 + 1] = {(v[9].re + v[9].im) * rh, (v[9].im - v[9].re) * rh};
  v[4 * 3 + 1] = cmul(v[13], cx<T>{s1, -c1});
  v[4 * 1 + 2] = {(v[6].re + v[6].im) * rh, (v[6].im - v[6].re) * rh};
  v[4 * 2 + 2] = {v[10].im, -v[10].re};
  v[4 * 3 + 2] = {(v[14].im - v[14].re) * rh, -(v[14].re + v[14].im) * rh};
  v[4 * 1 + 3] = cmul(v[7], cx<T>{s1, -c1});
  v[4 * 2 + 3] = {(v[11].im - v[11].re) * rh, -(v[11].re + v[11].im) * rh};
  v[4 * 3 + 3] = cmul(v[15], cx<T>{-c1, s1});
AFX_FFT_UNROLL
  for (int c = 0; c < 4; ++c) radix4(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
  // v[4c + d] = X[c + 4d]: transpose the 4x4 register grid (pure renaming)
  cx<T> t[16];
AFX_FFT_UNROLL
  for (int c = 0; c < 4; ++c)
AFX_FFT_UNROLL
    for (int d = 0; d < 4; ++d) t[c + 4 * d] = v[4 * c + d];
AFX_FFT_UNROLL
  for (int i = 0; i < 16; ++i) v[i] = t[i];
}

#if defined(__HIPCC__)

// E1/E2 through the wave's plane.  complex<float> is one 8-byte slot; complex<double> goes in two
// passes (real parts, then imaginary parts) through the same plane.
// ds_read_b64 with the register part of the address as an immediate.  Written as asm so that the
// load/store optimizer cannot fuse neighbours into ds_read2_b64 (half the LDS rate per byte on
// gfx950); the caller waits with lds_wait16 before using the values.
template <int BYTE_OFF>
__device__ __forceinline__ double lds_read_b64(unsigned addr) {
  double d;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(BYTE_OFF));
  return d;
}
__device__ __forceinline__ void lds_wait16(double (&d)[16]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]),
                 "+v"(d[8]), "+v"(d[9]), "+v"(d[10]), "+v"(d[11]), "+v"(d[12]), "+v"(d[13]), "+v"(d[14]), "+v"(d[15]));
}
template <int G>
__device__ __forceinline__ void lds_read16_step(unsigned addr, double (&d)[16]) {
  d[G] = lds_read_b64<8 * 65 * G>(addr);
  if constexpr (G + 1 < 16) lds_read16_step<G + 1>(addr, d);
}
// d[n2] = plane[lane + 65 n2], n2 = 0..15
__device__ __forceinline__ void lds_read16(unsigned addr, double (&d)[16]) {
  lds_read16_step<0>(addr, d);
  lds_wait16(d);
}

// E2: v[4 j2 + jl] of lane (jh, n2)  ->  v[n2] of lane k1 = 16 j2 + 4 jh + jl
template <typename T>
struct Xchg;
template <>
struct Xchg<float> {
  static __device__ __forceinline__ void run(unsigned char* plane, unsigned rd_addr, int wlane, cx<float> (&v)[16]) {
    float2* p = reinterpret_cast<float2*>(plane) + wlane;
    wave_lds_fence();
#pragma unroll
    for (int g = 0; g < 16; ++g) p[16 * (g >> 2) + (g & 3)] = make_float2(v[g].re, v[g].im);
    wave_lds_fence();
    double d[16];
    lds_read16(rd_addr, d);
    wave_lds_fence();
#pragma unroll
    for (int g = 0; g < 16; ++g)
      v[g] = {__int_as_float(__double2loint(d[g])), __int_as_float(__double2hiint(d[g]))};
  }
};
template <>
struct Xchg<double> {
  static __device__ __forceinline__ void run(unsigned char* plane, unsigned rd_addr, int wlane, cx<double> (&v)[16]) {
    double* p = reinterpret_cast<double*>(plane) + wlane;
    double re[16], im[16];
    wave_lds_fence();
#pragma unroll
    for (int g = 0; g < 16; ++g) p[16 * (g >> 2) + (g & 3)] = v[g].re;
    wave_lds_fence();
    lds_read16(rd_addr, re);
    wave_lds_fence();
#pragma unroll
    for (int g = 0; g < 16; ++g) p[16 * (g >> 2) + (g & 3)] = v[g].im;
    wave_lds_fence();
    lds_read16(rd_addr, im);
    wave_lds_fence();
#pragma unroll
    for (int g = 0; g < 16; ++g) v[g] = {re[g], im[g]};
  }
};

// E1 in registers: 2x2 block transposes between a lane bit and a register bit.
//   swap32(v[g], v[g+8]):  lane bit 5 <-> register bit 3      swap16(v[g], v[g+4]):  lane bit 4 <-> register bit 2
template <typename T>
__device__ __forceinline__ void reg_swap32(T& x, T& y) { swap32(x, y); }
template <typename T>
__device__ __forceinline__ void reg_swap16(T& x, T& y) { swap16(x, y); }
template <typename T>
__device__ __forceinline__ void transpose_m2_into_registers(cx<T> (&v)[16]) {
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    reg_swap32(v[g].re, v[g + 8].re);
    reg_swap32(v[g].im, v[g + 8].im);
  }
#pragma unroll
  for (int g = 0; g < 16; ++g)
    if ((g & 4) == 0) {
      reg_swap16(v[g].re, v[g + 4].re);
      reg_swap16(v[g].im, v[g + 4].im);
    }
}

// lane parts of the E2 addresses: this lane holds (jh, n2) when writing and is k1 when reading
__device__ __forceinline__ int fft_e2_write_slot(int lane) { return 4 * (lane >> 4) + 65 * (lane & 15); }

// Forward (e^{-i}) 1024-point FFT of v[r] = z[lane + 64 r]; returns v[k2] = Z[lane + 64 k2].
//   t1: global [4][4][4] table w64^(m2 (4 jh + jl)) offset to this lane's row (+ 16 (lane >> 4))
//   t2: LDS [16][64] table w1024^(n2 (j1 + 16 j2)) offset to this lane (+ lane)
//   plane / plane_rd_addr: the wave's exchange plane (pointer, and LDS byte address + 8 lane)
template <typename T>
__device__ __forceinline__ void fft1024(cx<T> (&v)[16], const cx<T>* t1, const cx<T>* t2, unsigned char* plane,
                                        unsigned plane_rd_addr, int lane) {
  dft16(v);
  transpose_m2_into_registers(v);   // lane = 16 jh + n2, v[4 m2 + jl]
#pragma unroll
  for (int g = 4; g < 16; ++g) v[g] = cmul(v[g], t1[g]);
#pragma unroll
  for (int jl = 0; jl < 4; ++jl) radix4(v[jl], v[4 + jl], v[8 + jl], v[12 + jl]);
#pragma unroll
  for (int g = 0; g < 16; ++g) v[g] = cmul(v[g], t2[64 * g]);
  Xchg<T>::run(plane, plane_rd_addr, fft_e2_write_slot(lane), v);
  dft16(v);
}

#endif  // __HIPCC__ (cross-lane stages)

}  // namespace
}  // namespace afx
