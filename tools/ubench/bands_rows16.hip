// tools/ubench/bands_rows16.hip -- the per-band sums of bands_kernel in two layouts, as a micro-benchmark (round 6: the
// "one frame per 16 lanes" layout the round-5 review asked to be budgeted before it is built).
//
//   V0  the shipped layout: one frame per 64-lane wave, bin k = 64 r + lane (12 rows in registers); a per-band sum is a
//       masked accumulation over the 25 (band, row) pairs into 16 accumulators + the transposed reduction wave_sum16
//       (swap32 / swap16 / DPP) -- afx_bands.hip: band_sum.
//   V1  four frames per wave, quad-interleaved: lane L = 4 m + f holds bins m + 16 i (i = 0 .. 47) of frame f.  Every
//       instruction of the accumulation works on four frames; the reduction over a frame's 16 lanes runs over lane bits
//       5, 4 (v_permlane32_swap, v_permlane16_swap: no selects), 3 (row_ror:8) and 2 (two bank-masked row shifts) and
//       leaves band m of frame f in lane 4 m + f -- where the closed forms then run in all 64 lanes, no parking in LDS.
//
// Both variants load their magnitudes from global memory into registers (V1: 48 doubles per lane) and form five sums
// per frame (x, x^2, x y and two selected sums, like sx / sxx / sxy / vsum / psum).  Output: ns per frame at a full chip
// (2 waves per SIMD, like bands_kernel), and how far the two layouts' sums agree.
// Build + run: hipcc --offload-arch=gfx950 -O3 -o /tmp/bands_rows16 tools/ubench/bands_rows16.hip && /tmp/bands_rows16
// Instruction counts: hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only (tools/isa_budget.py counts between markers).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

using mask64 = unsigned long long;
constexpr int kNumSub = 14;
constexpr int kSubStart[kNumSub + 1] = {1, 3, 7, 13, 23, 35, 50, 67, 90, 119, 160, 221, 317, 465, 752};

// hipcc of ROCm 7.2 materialises a 64-bit constant for an "s" operand of inline asm as `s_mov_b64 sN, <32-bit literal>`
// whenever the value fits a SIGN-extended int32, and the hardware ZERO-extends that literal: a lane mask like
// 0xfffffffff0000000 arrives as 0x00000000f0000000 (seen in this file's own first version: lanes 32..63 lost).  Such a
// mask is passed as its complement, which is a small positive value, with the two sources of the select swapped.
__device__ __forceinline__ constexpr bool literal_would_be_zero_extended(mask64 m) {
  return (long long)m < -16 && (long long)m >= -(1ll << 31);
}
__device__ __forceinline__ double keep_where(double v, mask64 m) {
  int lo, hi;
  if (literal_would_be_zero_extended(m)) {
    asm("v_cndmask_b32_e64 %0, %1, 0, %2" : "=v"(lo) : "v"(__double2loint(v)), "s"(~m));
    asm("v_cndmask_b32_e64 %0, %1, 0, %2" : "=v"(hi) : "v"(__double2hiint(v)), "s"(~m));
  } else {
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(lo) : "v"(__double2loint(v)), "s"(m));
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(hi) : "v"(__double2hiint(v)), "s"(m));
  }
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void swap32(double& x, double& y) {
  const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(x), __double2loint(y), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(x), __double2hiint(y), false, false);
  x = __hiloint2double(hi[0], lo[0]);
  y = __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ void swap16(double& x, double& y) {
  const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(x), __double2loint(y), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(x), __double2hiint(y), false, false);
  x = __hiloint2double(hi[0], lo[0]);
  y = __hiloint2double(hi[1], lo[1]);
}
// value of lane ^ 4 inside the row: two bank-masked row shifts (banks = groups of four lanes)
__device__ __forceinline__ double xor4(double v) {
  int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x104 /* row_shl:4 */, 0xF, 0x5, false);
  lo = __builtin_amdgcn_update_dpp(lo, __double2loint(v), 0x114 /* row_shr:4 */, 0xF, 0xA, false);
  int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x104, 0xF, 0x5, false);
  hi = __builtin_amdgcn_update_dpp(hi, __double2hiint(v), 0x114, 0xF, 0xA, false);
  return __hiloint2double(hi, lo);
}

// ---- V0: the shipped layout ----
constexpr int kRows0 = 12;
constexpr bool touches0(int b, int r) { return kSubStart[b] <= 64 * r + 63 && kSubStart[b + 1] - 1 >= 64 * r; }
constexpr bool covers0(int b, int r) { return kSubStart[b] <= 64 * r && kSubStart[b + 1] - 1 >= 64 * r + 63; }
constexpr mask64 mask0(int b, int r) {
  const int lo = kSubStart[b], hi = kSubStart[b + 1];
  const int l0 = lo - 64 * r < 0 ? 0 : (lo - 64 * r > 64 ? 64 : lo - 64 * r), l1 = hi - 64 * r < 0 ? 0 : (hi - 64 * r > 64 ? 64 : hi - 64 * r);
  const mask64 a = l1 >= 64 ? ~0ull : ((1ull << l1) - 1ull), c = l0 >= 64 ? ~0ull : ((1ull << l0) - 1ull);
  return l1 > l0 ? (a & ~c) : 0ull;
}
struct T0 { mask64 m[kNumSub][kRows0]; };
constexpr T0 make0() { T0 t{}; for (int b = 0; b < kNumSub; ++b) for (int r = 0; r < kRows0; ++r) t.m[b][r] = mask0(b, r); return t; }
constexpr T0 kM0 = make0();
__device__ __forceinline__ double wave_sum16(double (&a)[16], int lane) {
#pragma unroll
  for (int i = 0; i < 8; ++i) { swap32(a[i], a[i + 8]); a[i] += a[i + 8]; }
#pragma unroll
  for (int i = 0; i < 4; ++i) { swap16(a[i], a[i + 4]); a[i] += a[i + 4]; }
  const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const double keep = b3 ? a[i + 2] : a[i], send = b3 ? a[i] : a[i + 2];
    a[i] = keep + dpp_mov<0x128>(send);
  }
  const double keep = b2 ? a[1] : a[0], send = b2 ? a[0] : a[1];
  double z = keep + dpp_mov<0x141>(send);
  z += dpp_mov<0x4E>(z);
  z += dpp_mov<0xB1>(z);
  return z;
}
template <typename F>
__device__ __forceinline__ double band_sum0(F value_of_row, int lane) {
  double acc[16];
#pragma unroll
  for (int b = 0; b < 16; ++b) {
    acc[b] = 0.0;
    if (b < kNumSub) {
      bool first = true;
#pragma unroll
      for (int r = 0; r < kRows0; ++r)
        if (touches0(b, r)) {
          const double v = covers0(b, r) ? value_of_row(r) : keep_where(value_of_row(r), kM0.m[b][r]);
          acc[b] = first ? v : acc[b] + v;
          first = false;
        }
    }
  }
  return wave_sum16(acc, lane);
}

// ---- V1: four frames per wave, lane = 4 m + f, bin = m + 16 i ----
constexpr int kRows1 = 48;
constexpr bool touches1(int b, int i) { return kSubStart[b] <= 16 * i + 15 && kSubStart[b + 1] - 1 >= 16 * i; }
constexpr bool covers1(int b, int i) { return kSubStart[b] <= 16 * i && kSubStart[b + 1] - 1 >= 16 * i + 15; }
constexpr mask64 mask1(int b, int i) {   // lanes 4 m + f with bin m + 16 i inside band b, every f
  mask64 m = 0;
  for (int q = 0; q < 16; ++q)
    if (q + 16 * i >= kSubStart[b] && q + 16 * i < kSubStart[b + 1]) m |= 0xFull << (4 * q);
  return m;
}
struct T1 { mask64 m[kNumSub][kRows1]; };
constexpr T1 make1() { T1 t{}; for (int b = 0; b < kNumSub; ++b) for (int i = 0; i < kRows1; ++i) t.m[b][i] = mask1(b, i); return t; }
constexpr T1 kM1 = make1();
// 16 accumulators over the 16 lanes of each frame: lane 4 m + f receives the total of a[m] of frame f
__device__ __forceinline__ double row_sum16(double (&a)[16], int lane) {
#pragma unroll
  for (int i = 0; i < 8; ++i) { swap32(a[i], a[i + 8]); a[i] += a[i + 8]; }     // m bit 3 = lane bit 5
#pragma unroll
  for (int i = 0; i < 4; ++i) { swap16(a[i], a[i + 4]); a[i] += a[i + 4]; }     // m bit 2 = lane bit 4
  const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {                                                  // m bit 1 = lane bit 3
    const double keep = b3 ? a[i + 2] : a[i], send = b3 ? a[i] : a[i + 2];
    a[i] = keep + dpp_mov<0x128>(send);
  }
  const double keep = b2 ? a[1] : a[0], send = b2 ? a[0] : a[1];                 // m bit 0 = lane bit 2
  return keep + xor4(send);
}
constexpr int first_row1(int b) { for (int i = 0; i < kRows1; ++i) if (touches1(b, i)) return i; return -1; }
// rows as template recursion: a 14 x 48 loop nest under "#pragma unroll" was left as loops with run-time register indexing
// (s_set_gpr_idx) and the masks in a constant table in memory
template <int I, typename F>
__device__ __forceinline__ void accumulate_row1(double (&acc)[16], F& value_of_row) {
  if constexpr (I < kRows1) {
    const double v = value_of_row(I);
#pragma unroll
    for (int b = 0; b < kNumSub; ++b)
      if (touches1(b, I)) {
        const double w = covers1(b, I) ? v : keep_where(v, kM1.m[b][I]);
        acc[b] = (first_row1(b) == I) ? w : acc[b] + w;
      }
    accumulate_row1<I + 1>(acc, value_of_row);
  }
}
template <typename F>
__device__ __forceinline__ double band_sum1(F value_of_row, int lane) {
  double acc[16];
  acc[14] = 0.0; acc[15] = 0.0;
  accumulate_row1<0>(acc, value_of_row);
  return row_sum16(acc, lane);
}

constexpr int kImage = 768;             // doubles per frame
constexpr int kWindow = 16;             // frames a wave cycles through

// (Round 6, first attempt: V1 read its four frames from LDS images like the shipped kernel reads its one -- 4 x 6 KiB per
// wave, 5 with the predecessor: 8 waves per CU would need 240 KB of the 160 KB, i.e. ONE wave per SIMD.  The layout only
// works with the spectrum in registers, loaded from global memory: 48 doubles per lane.)
template <int V>
__global__ __launch_bounds__(256, 2) void bench_kernel(const double* spectra, int frames_per_wave, double* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gw = blockIdx.x * 4 + wave;
  double total = 0.0;
  // every wave walks frames_per_wave frames of a window of kWindow frames shared by 64 waves (6 MB in all: the spectra come
  // out of the cache, as in the pipeline where the band kernel is VALU-bound -- streamed from HBM both variants measured
  // the memory system, 3.2 TB/s)
  const double* src = spectra + (size_t)(gw & 63) * kWindow * kImage;
  if (V == 0) {
    double y[kRows0];
#pragma unroll
    for (int r = 0; r < kRows0; ++r) y[r] = 0.5;
    for (int f = 0; f < frames_per_wave; ++f) {
      double x[kRows0], xx[kRows0], xy[kRows0], xs[kRows0], xp[kRows0];
#pragma unroll
      for (int r = 0; r < kRows0; ++r) x[r] = src[(size_t)(f & (kWindow - 1)) * kImage + 64 * r + lane];
#pragma unroll
      for (int r = 0; r < kRows0; ++r) {
        xx[r] = x[r] * x[r]; xy[r] = x[r] * y[r];
        xs[r] = x[r] < 0.3 ? x[r] : 0.0; xp[r] = x[r] > 0.7 ? x[r] : 0.0;
      }
      const double s0 = band_sum0([&](int r) { return x[r]; }, lane);
      __builtin_amdgcn_sched_barrier(0);
      const double s1 = band_sum0([&](int r) { return xx[r]; }, lane);
      __builtin_amdgcn_sched_barrier(0);
      const double s2 = band_sum0([&](int r) { return xy[r]; }, lane);
      __builtin_amdgcn_sched_barrier(0);
      const double s3 = band_sum0([&](int r) { return xs[r]; }, lane);
      __builtin_amdgcn_sched_barrier(0);
      const double s4 = band_sum0([&](int r) { return xp[r]; }, lane);
      total += ((lane & 3) == 0 && (lane >> 2) < kNumSub) ? s0 + 2 * s1 + 5 * s3 + 7 * s4 + 1e-30 * s2 : 0.0;
#pragma unroll
      for (int r = 0; r < kRows0; ++r) y[r] = x[r];
    }
  } else {
    const int m = lane >> 2, fq = lane & 3;
    for (int f = 0; f + 3 < frames_per_wave; f += 4) {
      const double* mine = src + (size_t)((f + fq) & (kWindow - 1)) * kImage + m;
      const double* prev = src + (size_t)((f + ((fq + 3) & 3)) & (kWindow - 1)) * kImage + m;     // stand-in for the frame before
      double x[kRows1];
#pragma unroll
      for (int i = 0; i < kRows1; ++i) x[i] = mine[16 * i];
      const double s0 = band_sum1([&](int i) { return x[i]; }, lane);
      __builtin_amdgcn_sched_barrier(0);
      const double s1 = band_sum1([&](int i) { return x[i] * x[i]; }, lane);
      __builtin_amdgcn_sched_barrier(0);
      const double s2 = band_sum1([&](int i) { return x[i] * prev[16 * i]; }, lane);
      __builtin_amdgcn_sched_barrier(0);
      const double s3 = band_sum1([&](int i) { return x[i] < 0.3 ? x[i] : 0.0; }, lane);
      __builtin_amdgcn_sched_barrier(0);
      const double s4 = band_sum1([&](int i) { return x[i] > 0.7 ? x[i] : 0.0; }, lane);
      total += (m < kNumSub) ? s0 + 2 * s1 + 5 * s3 + 7 * s4 : 0.0;
      // (the x y sums of V1 pair a frame with another one than V0 does: left out of the checksum, kept in the timing)
      total += (m < kNumSub) ? 1e-30 * s2 : 0.0;
    }
  }
#pragma unroll
  for (int s = 32; s; s >>= 1) total += __shfl_xor(total, s);
  if (lane == 0) out[gw] = total;
}

// self-check: sum of x per band of four frames, both layouts, against the host (one wave each)
__global__ void check_kernel(const double* spectra, double* out) {
  const int lane = threadIdx.x & 63;
  for (int f = 0; f < 4; ++f) {      // V0: one frame at a time
    double x[kRows0];
#pragma unroll
    for (int r = 0; r < kRows0; ++r) x[r] = spectra[(size_t)f * kImage + 64 * r + lane];
    const double s0 = band_sum0([&](int r) { return x[r]; }, lane);
    if ((lane & 3) == 0) out[16 * f + (lane >> 2)] = s0;
  }
  {                                   // V1: the four at once
    const int m = lane >> 2, fq = lane & 3;
    double x[kRows1];
#pragma unroll
    for (int i = 0; i < kRows1; ++i) x[i] = spectra[(size_t)fq * kImage + m + 16 * i];
    const double s0 = band_sum1([&](int i) { return x[i]; }, lane);
    out[64 + 16 * fq + m] = s0;
  }
}

int main() {
  const int blocks = 512, waves = blocks * 4, fpw = 256;   // two workgroups of four waves per CU: 2 waves per SIMD, like bands_kernel
  const size_t n = (size_t)64 * kWindow * kImage;
  std::vector<double> h(n);
  unsigned s = 12345;
  for (double& v : h) { s = s * 1664525u + 1013904223u; v = (double)(s >> 8) / 16777216.0; }
  double *d, *o;
  hipMalloc(&d, n * 8); hipMalloc(&o, waves * 8);
  hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice);
  {
    double* dc; hipMalloc(&dc, 128 * 8);
    hipLaunchKernelGGL(check_kernel, dim3(1), dim3(64), 0, 0, d, dc);
    double hc[128];
    hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost);
    double worst0 = 0, worst1 = 0;
    for (int f = 0; f < 4; ++f)
      for (int b = 0; b < kNumSub; ++b) {
        double want = 0;
        for (int k = kSubStart[b]; k < kSubStart[b + 1]; ++k) want += h[(size_t)f * kImage + k];
        worst0 = std::fmax(worst0, std::fabs(hc[16 * f + b] - want) / want);
        worst1 = std::fmax(worst1, std::fabs(hc[64 + 16 * f + b] - want) / want);
      }
    std::printf("self-check, per-band sums of four frames against the host: V0 worst rel. err %.2e, V1 %.2e\n", worst0, worst1);
  }
  double checksum[2] = {0, 0};
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int v = 0; v < 2; ++v) {
    float best = 1e30f;
    std::vector<double> ho(waves);
    for (int rep = 0; rep < 12; ++rep) {
      hipEventRecord(e0);
      if (v == 0) hipLaunchKernelGGL(bench_kernel<0>, dim3(blocks), dim3(256), 0, 0, d, fpw, o);
      else hipLaunchKernelGGL(bench_kernel<1>, dim3(blocks), dim3(256), 0, 0, d, fpw, o);
      if (hipGetLastError() != hipSuccess) { std::printf("launch failed\n"); return 1; }
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep >= 4 && ms < best) best = ms;
    }
    hipMemcpy(ho.data(), o, waves * 8, hipMemcpyDeviceToHost);
    double sum = 0; for (double q : ho) sum += q;
    checksum[v] = sum;      // sum over frames and bands of s(x) + 2 s(x^2) + 5 s(x < 0.3) + 7 s(x > 0.7): the same in both layouts
    const double frames = (double)waves * fpw;
    std::printf("V%d %-44s %8.3f ms for %.0f frames x 5 band sums = %7.2f ns per frame at a full chip, %.1f ps per (frame, sum)\n", v,
                v == 0 ? "one frame per wave (shipped layout)" : "four frames per wave, lane = 4 m + f", best, frames, best * 1e6 / frames,
                best * 1e9 / frames / 5);
  }
  std::printf("the two layouts' sums agree to %.2e relative (summation orders differ)\n", std::fabs(checksum[0] - checksum[1]) / std::fabs(checksum[0]));
  return 0;
}
