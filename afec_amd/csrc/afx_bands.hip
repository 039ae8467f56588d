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
//   contrast                     one 1024-slot bitonic sort in registers on the 32-bit key
//                                (band << 27 | float bits >> 4); after the sort every band
//                                occupies a fixed range of positions, so "mean of the n lowest /
//                                highest" (std::sort + two loops, SA:2200-2228) is a sum over
//                                fixed position ranges of the LDS-staged sorted values

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

// ---- 1024-slot bitonic sort of 32-bit keys, p = 16 lane + reg, every comparator ascending ----
// ("flip" form: the first stage of a merge of size K pairs p with p ^ (K-1), the rest with p ^ J).
// In-lane comparators are v_min_u32 / v_max_u32; cross-lane partners come through DPP where the
// lane permutation is one (xor 1, 2, 8; mirrors 3, 7, 15) and through ds_bpermute otherwise.
using u32 = unsigned;

template <int M>   // partner = lane ^ M for M in {1, 2, 3, 4, 7, 8, 15, 16, 31, 32, 63}
__device__ __forceinline__ u32 lane_xor_u32(u32 v) {
  if constexpr (M == 1) return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);        // quad_perm [1,0,3,2]
  else if constexpr (M == 2) return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
  else if constexpr (M == 3) return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x1B, 0xF, 0xF, true);   // quad_perm [3,2,1,0]
  else if constexpr (M == 7) return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);  // row_half_mirror
  else if constexpr (M == 8) return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, true);  // row_ror:8
  else if constexpr (M == 15) return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true); // row_mirror
  else return (u32)__shfl_xor((int)v, M);
}

template <int J>   // in-lane stage: pairs (i, i ^ J) for J < 16, or the flip (i, i ^ (K-1)) for K <= 16 via MASK
__device__ __forceinline__ void inlane_stage(u32 (&key)[16]) {
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int l = i ^ J;
    if (l > i) {
      const u32 a = key[i], b = key[l];
      key[i] = a < b ? a : b;
      key[l] = a < b ? b : a;
    }
  }
}
// cross-lane stage: partner lane = lane ^ M, partner register = reg ^ RX (RX = 15 for a flip, 0 otherwise)
template <int M, int RX>
__device__ __forceinline__ void crosslane_stage(u32 (&key)[16], int lane) {
  constexpr int TOP = (M + 1) >> 1 > 0 ? ((M & (M + 1)) == 0 ? (M + 1) >> 1 : M) : M;  // highest set bit of M
  const bool lower = (lane & TOP) == 0;
  u32 out[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const u32 a = key[i];
    const u32 p = lane_xor_u32<M>(key[i ^ RX]);
    const u32 lo = a < p ? a : p, hi = a < p ? p : a;
    out[i] = lower ? lo : hi;
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) key[i] = out[i];
}
template <int K, int J>
__device__ __forceinline__ void merge_tail(u32 (&key)[16], int lane) {
  if constexpr (J >= 16) crosslane_stage<(J >> 4), 0>(key, lane);
  else inlane_stage<J>(key);
  if constexpr (J > 1) merge_tail<K, J / 2>(key, lane);
}
template <int K>
__device__ __forceinline__ void sort_level(u32 (&key)[16], int lane) {
  if constexpr (K > 2) sort_level<K / 2>(key, lane);
  if constexpr (K <= 16) inlane_stage<K - 1>(key);          // flip inside the lane
  else crosslane_stage<(K >> 4) - 1, 15>(key, lane);        // flip across lanes: lane ^ (K/16 - 1), reg ^ 15
  if constexpr (K >= 4) merge_tail<K, K / 4>(key, lane);
}

__global__ __launch_bounds__(256, 2) void bands_kernel(const BandArgs a) {
  const int lane0 = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t stride = (int64_t)gridDim.x * 4;
  __shared__ double s_thr[4][16];
  __shared__ float s_sorted[4][1024];
  double* const thr = s_thr[threadIdx.x >> 6];
  float* const sorted = s_sorted[threadIdx.x >> 6];

  for (int64_t f = wave0; f < a.n_frames; f += stride) {
    // re-materialise the lane id every frame: keeps the hundreds of lane-range predicates below from
    // being hoisted out of the loop into (spilled) SGPR pairs
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const double* const cur = a.mag + f * kHalf;
    const double* const prv = a.mag + (int64_t)a.prev[f] * kHalf;
    double x[kRows], y[kRows];
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
      x[r] = cur[64 * r + lane];
      y[r] = prv[64 * r + lane];
    }

    // ---- spectral_flux: Pearson r with the previous frame over bins 1..738 (SA:1919-1933,
    //      Statistics.cpp:604-638); frame 0 of a buffer is compared with itself (SA:937-940) ----
    if (a.flags & kBandsFlux) {
      double fa = 0.0, fb = 0.0, faa = 0.0, fbb = 0.0, fab = 0.0;
#pragma unroll
      for (int r = 0; r < kRows; ++r) {
        const int k = 64 * r + lane;
        const bool ok = k >= kFirstBin && k <= kLastBin;
        const double p = ok ? x[r] : 0.0, q = ok ? y[r] : 0.0;
        fa += p; fb += q; faa += p * p; fbb += q * q; fab += p * q;
      }
      fa = wave_sum(fa); fb = wave_sum(fb); faa = wave_sum(faa); fbb = wave_sum(fbb); fab = wave_sum(fab);
      const double n = (double)kBinCount;
      const double ma = fa / n, mb = fb / n;
      const double denom2 = (faa - ma * ma * n) * (fbb - mb * mb * n);
      const double num = fab - (ma * mb * n);
      if (lane == 0) a.rec[f * a.lay.stride + a.lay.flux] = (fabs(denom2) > (double)1e-12f) ? num / sqrt(denom2) : 0.0;
    }
    if (!(a.flags & kBandsFeatures)) continue;

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

    // ---- contrast: sort (band, value) keys, then fixed position ranges ----
    // key = band << 27 | float bits >> 4 (rounded): 19 mantissa bits order the values, and the sums
    // below are rebuilt from the keys (relative error <= 2^-20 per element, far inside the 1e-4 bar)
    u32 key[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (r < kRows) {
        int bid = 15;
#pragma unroll
        for (int b = 0; b < kNumSub; ++b)
          if (sub_touches(b, r)) bid = in_band(b, r, lane) ? b : bid;
        const u32 bits = __float_as_uint((float)fabs(x[r]));
        key[r] = ((u32)bid << 27) | ((bits + 8u) >> 4);
      } else {
        key[r] = 0x7FFFFFFFu;
      }
    }
    sort_level<1024>(key, lane);
    // sorted position p = 16 lane + i; stage the values in LDS in position order
    wave_lds_fence();
#pragma unroll
    for (int i = 0; i < 16; ++i) sorted[16 * lane + i] = __uint_as_float((key[i] & 0x07FFFFFFu) << 4);
    wave_lds_fence();
    // band b sits at [pos0, pos0 + n): valley = first nn, peak = last nn (nn <= 86 -> two passes of 64 lanes)
    double valley_acc[16], peak_acc[16];
#pragma unroll
    for (int b = 0; b < 16; ++b) {
      valley_acc[b] = 0.0;
      peak_acc[b] = 0.0;
      if (b < kNumSub) {
        const int lo = sub_pos0(b), n = kSubN[b], nn = kSubNeigh[b];
#pragma unroll
        for (int t0 = 0; t0 < nn; t0 += 64) {
          const int t = t0 + lane;
          valley_acc[b] += (t < nn) ? (double)sorted[lo + (t < nn ? t : 0)] : 0.0;
          peak_acc[b] += (t < nn) ? (double)sorted[lo + n - nn + (t < nn ? t : 0)] : 0.0;
        }
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
    const double gm = fast_exp(slog / nb);                             // Statistics.cpp:442
    const double fl = (mean == 0.0) ? 0.0 : gm / mean;                 // Statistics.cpp:565-574
    double fdb = lin_to_db(fl) / -60.0;                                // SFlatnessDb, SA:129-133
    fdb = fdb < 1.0 ? fdb : 1.0;
    const double ma = sx / nb, mb = sy / nb;                           // Statistics.cpp:604-638
    const double denom2 = (sxx - ma * ma * nb) * (syy - mb * mb * nb);
    const double num = sxy - (ma * mb * nb);
    const double flux = (fabs(denom2) > (double)1e-12f) ? num / sqrt(denom2) : 0.0;
    const double valley = vsum / nn + 1e-30, peakv = psum / nn + 1e-30;  // SA:2216, 2228
    // pow(a, b) = exp(b log a), a > 0 (SA:2231-2232)
    const double contrast = -1.0 * fast_exp(fast_log(peakv / valley) / fast_log(mean + 1e-30));

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
