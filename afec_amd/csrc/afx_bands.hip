// afec_amd/csrc/afx_bands.hip -- the 14 "sub-band" descriptors of
// TSampleAnalyser::CalcSpectralBandFeatures (SampleAnalyser.cpp:2067-2308) on gfx950.
//
// One wave per frame, working from the magnitude spectrum the frame kernel left in HBM
// ([F][1024] doubles, L2/MALL-resident between the two launches for realistic batch sizes) and
// the previous frame's spectrum of the same buffer (flux).  Bands are laid out contiguously from
// bin 1 with the reference's bin counts (the drift from the nominal edges is intentional, see
// SURVEY 8a/a15), so every membership test below is a compile-time range of k = 64 r + lane.
//
//   rms, flatness(dB), flux      masked per-band sums, reduced 16 bands at a time (wave_sum16)
//   complexity                   band maximum (wave_max16) -> threshold -> strict local maxima
//   contrast                     one 1024-slot bitonic sort in registers on the key
//                                (band << 60 | double bits >> 4); after the sort every band
//                                occupies a fixed range of positions, so "mean of the n lowest /
//                                highest" (std::sort + two loops, SA:2200-2228) is again a
//                                masked sum over static positions

#include <hip/hip_runtime.h>

#include "afx_internal.h"
#include "afx_device.h"

namespace afx {
namespace {

// contiguous layout from bin 1 with the counts kSubN (afx_internal.h)
constexpr int kSubStart[kNumSub + 1] = {1, 3, 7, 13, 23, 35, 50, 67, 90, 119, 160, 221, 317, 465, 752};
// max(1, (int)(0.3 * n)), SA:2204
constexpr int kSubNeigh[kNumSub] = {1, 1, 1, 3, 3, 4, 5, 6, 8, 12, 18, 28, 44, 86};
constexpr int kRows = 12;  // bins 0..767

constexpr bool sub_touches(int b, int r) { return kSubStart[b] <= 64 * r + 63 && kSubStart[b + 1] - 1 >= 64 * r; }
// position range of band b in the sorted array (bands are sorted by id first)
constexpr int sub_pos0(int b) { return kSubStart[b] - 1; }
constexpr bool pos_touches(int lo, int hi, int lane_lo) { return lo <= lane_lo + 15 && hi - 1 >= lane_lo; }

__device__ __forceinline__ bool in_band(int b, int r, int lane) {
  const int k = 64 * r + lane;
  return k >= kSubStart[b] && k < kSubStart[b + 1];
}

// per-band sum of one value per row; lane L receives the total of band (L >> 2) & 15
template <typename F>
__device__ __forceinline__ double band_sum(F value_of_row, int lane) {
  double acc[16];
#pragma unroll
  for (int b = 0; b < 16; ++b) {
    acc[b] = 0.0;
    if (b < kNumSub) {
#pragma unroll
      for (int r = 0; r < kRows; ++r)
        if (sub_touches(b, r)) acc[b] += in_band(b, r, lane) ? value_of_row(r) : 0.0;
    }
  }
  return wave_sum16(acc, lane);
}
template <typename F>
__device__ __forceinline__ double band_max(F value_of_row, int lane) {
  double acc[16];
#pragma unroll
  for (int b = 0; b < 16; ++b) {
    acc[b] = 0.0;
    if (b < kNumSub) {
#pragma unroll
      for (int r = 0; r < kRows; ++r)
        if (sub_touches(b, r)) acc[b] = fmax(acc[b], in_band(b, r, lane) ? value_of_row(r) : 0.0);
    }
  }
  return wave_max16(acc, lane);
}

using u64 = unsigned long long;

template <int CTRL>
__device__ __forceinline__ u64 dpp_mov_u64(u64 v) {
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)v, CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, 0xF, 0xF, true);
  return ((u64)(unsigned)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ u64 shfl_xor_u64(u64 v, int m) {
  const int lo = __shfl_xor((int)(unsigned)v, m);
  const int hi = __shfl_xor((int)(unsigned)(v >> 32), m);
  return ((u64)(unsigned)hi << 32) | (unsigned)lo;
}
template <int LM>
__device__ __forceinline__ u64 lane_xor(u64 v) {
  if constexpr (LM == 1) return dpp_mov_u64<kDppXor1>(v);
  else if constexpr (LM == 2) return dpp_mov_u64<kDppXor2>(v);
  else if constexpr (LM == 8) return dpp_mov_u64<kDppRor8>(v);
  else return shfl_xor_u64(v, LM);
}

// one stage (distance J inside merges of size K) of the bitonic network over p = 16 lane + reg
template <int K, int J>
__device__ __forceinline__ void bitonic_stage(u64 (&key)[16], int lane) {
  if constexpr (J < 16) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if ((i & J) == 0) {
        const int l = i | J;
        bool asc;
        if constexpr (K <= 16) asc = (K == 16) ? ((lane & 1) == 0) : ((i & K) == 0);
        else asc = (lane & (K >> 4)) == 0;
        const u64 a = key[i], b = key[l];
        const bool swap = (a > b) == asc;
        key[i] = swap ? b : a;
        key[l] = swap ? a : b;
      }
    }
  } else {
    constexpr int LM = J >> 4;
    const bool asc = (K >= 1024) ? true : ((lane & (K >> 4)) == 0);
    const bool lower = (lane & LM) == 0;
    const bool take_min = (lower == asc);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const u64 a = key[i];
      const u64 p = lane_xor<LM>(a);
      const bool a_lt = a < p;
      key[i] = (a_lt == take_min) ? a : p;
    }
  }
}
template <int K, int J>
__device__ __forceinline__ void bitonic_merge(u64 (&key)[16], int lane) {
  bitonic_stage<K, J>(key, lane);
  if constexpr (J > 1) bitonic_merge<K, J / 2>(key, lane);
}
template <int K>
__device__ __forceinline__ void bitonic_sort(u64 (&key)[16], int lane) {
  if constexpr (K > 2) bitonic_sort<K / 2>(key, lane);
  bitonic_merge<K, K / 2>(key, lane);
}

__global__ __launch_bounds__(256) void bands_kernel(const BandArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t stride = (int64_t)gridDim.x * 4;
  __shared__ double s_thr[4][16];
  double* const thr = s_thr[threadIdx.x >> 6];

  for (int64_t f = wave0; f < a.n_frames; f += stride) {
    const double* const cur = a.mag + f * kHalf;
    const double* const prv = a.mag + (int64_t)a.prev[f] * kHalf;
    double x[kRows], y[kRows];
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
      x[r] = cur[64 * r + lane];
      y[r] = prv[64 * r + lane];
    }

    // ---- masked per-band sums: lane L ends up with band (L >> 2) & 15 ----
    const double sx = band_sum([&](int r) { return x[r]; }, lane);
    const double sxx = band_sum([&](int r) { return x[r] * x[r]; }, lane);
    const double sy = band_sum([&](int r) { return y[r]; }, lane);
    const double syy = band_sum([&](int r) { return y[r] * y[r]; }, lane);
    const double sxy = band_sum([&](int r) { return x[r] * y[r]; }, lane);
    // geometric mean: sum of log(|x| + 1e-20) (Statistics.cpp:417-455 keeps a running product and
    // takes logs only when it leaves [1e-64, 1e64]; same value up to rounding)
    double lg[kRows];
#pragma unroll
    for (int r = 0; r < kRows; ++r) lg[r] = fast_log(fabs(x[r]) + 1e-20);
    const double slog = band_sum([&](int r) { return lg[r]; }, lane);
    const double bmax = band_max([&](int r) { return x[r]; }, lane);

    // ---- complexity: strict local maxima above 0.25 * band maximum (SA:2170-2197) ----
    wave_lds_fence();
    if ((lane & 3) == 0) thr[lane >> 2] = bmax * 0.25;
    wave_lds_fence();
    double pk[kRows];
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
      const int k = 64 * r + lane;
      int bid = -1;
#pragma unroll
      for (int b = 0; b < kNumSub; ++b)
        if (sub_touches(b, r)) bid = in_band(b, r, lane) ? b : bid;
      const double t = thr[bid < 0 ? 15 : bid];
      // neighbours in the unsorted spectrum, may reach into the adjacent band; bin 0 and 1023 never count
      const double left = (k > 0) ? cur[k - 1] : 0.0;
      const double right = cur[k + 1];
      const bool peak = bid >= 0 && t > 0.0 && x[r] > t && k > 0 && x[r] > left && x[r] > right;
      pk[r] = peak ? 1.0 : 0.0;
    }
    const double cplx = band_sum([&](int r) { return pk[r]; }, lane);

    // ---- contrast: sort (band, value) keys, then static position ranges ----
    u64 key[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (r < kRows) {
        int bid = 15;
#pragma unroll
        for (int b = 0; b < kNumSub; ++b)
          if (sub_touches(b, r)) bid = in_band(b, r, lane) ? b : bid;
        const u64 bits = (u64)__double_as_longlong(fabs(x[r]));
        key[r] = ((u64)bid << 60) | (bits >> 4);
      } else {
        key[r] = ~0ull;
      }
    }
    bitonic_sort<1024>(key, lane);
    // sorted position p = 16 lane + i holds key[i]; band b sits at [sub_pos0(b), sub_pos0(b) + n_b)
    double valley_acc[16], peak_acc[16];
#pragma unroll
    for (int b = 0; b < 16; ++b) {
      valley_acc[b] = 0.0;
      peak_acc[b] = 0.0;
    }
    const int p0 = 16 * lane;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int p = p0 + i;
      const double v = __longlong_as_double((long long)((key[i] & 0x0FFFFFFFFFFFFFFFull) << 4));
#pragma unroll
      for (int b = 0; b < kNumSub; ++b) {
        const int lo = sub_pos0(b), n = kSubN[b], nn = kSubNeigh[b];
        valley_acc[b] += (p >= lo && p < lo + nn) ? v : 0.0;
        peak_acc[b] += (p >= lo + n - nn && p < lo + n) ? v : 0.0;
      }
    }
    const double vsum = wave_sum16(valley_acc, lane);
    const double psum = wave_sum16(peak_acc, lane);

    // ---- per-band results: every lane finishes the band (lane >> 2) & 15 ----
    const int b = (lane >> 2) & 15;
    const bool valid = b < kNumSub;
    const int bb = valid ? b : 0;
    double nb = 1.0, nn = 1.0;
#pragma unroll
    for (int i = 0; i < kNumSub; ++i) {
      nb = (bb == i) ? (double)kSubN[i] : nb;
      nn = (bb == i) ? (double)kSubNeigh[i] : nn;
    }
    const double mean = (nb >= 2.0) ? sx / nb : sx;                    // TStatistics::Mean
    const double rms = sqrt(sxx / nb);                                 // SA:2154-2159
    const double gm = exp(slog / nb);                                  // Statistics.cpp:442
    const double fl = (mean == 0.0) ? 0.0 : gm / mean;                 // Statistics.cpp:565-574
    double fdb = lin_to_db(fl) / -60.0;                                // SFlatnessDb, SA:129-133
    fdb = fdb < 1.0 ? fdb : 1.0;
    const double ma = sx / nb, mb = sy / nb;                           // Statistics.cpp:604-638
    const double denom2 = (sxx - ma * ma * nb) * (syy - mb * mb * nb);
    const double num = sxy - (ma * mb * nb);
    const double flux = (fabs(denom2) > (double)1e-12f) ? num / sqrt(denom2) : 0.0;
    const double valley = vsum / nn + 1e-30, peakv = psum / nn + 1e-30;  // SA:2216, 2228
    const double contrast = -1.0 * pow(peakv / valley, 1.0 / log(mean + 1e-30));  // SA:2231-2232

    double* const rec = a.rec + f * a.lay.stride;
    if (valid && (lane & 3) == 0) {
      rec[a.lay.sub_rms + b] = rms;
      rec[a.lay.sub_flat + b] = fdb;
      rec[a.lay.sub_flux + b] = flux;
      rec[a.lay.sub_cplx + b] = cplx;
      rec[a.lay.sub_contrast + b] = contrast;
    }
    // spectral_contrast = mean of the 14 band contrasts, summed in band order (SA:2252-2260)
    double csum = 0.0;
#pragma unroll
    for (int i = 0; i < kNumSub; ++i) csum += __shfl(contrast, 4 * i);
    if (lane == 0) rec[a.lay.contrast] = csum / (double)kNumSub;
  }
}

}  // namespace

hipError_t launch_bands(const BandArgs& a, hipStream_t stream) {
  if (a.n_frames <= 0) return hipSuccess;
  const int64_t want = (a.n_frames + 3) / 4;
  const int grid = (int)(want < 256 * 16 ? want : 256 * 16);
  hipLaunchKernelGGL(bands_kernel, dim3(grid), dim3(256), 0, stream, a);
  return hipGetLastError();
}

}  // namespace afx
