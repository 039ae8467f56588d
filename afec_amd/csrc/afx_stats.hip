// afec_amd/csrc/afx_stats.hip -- per-file statistics of every framed series:
// TStatistics::Calc (Statistics.cpp:12-90) as TFramedScalarData / TFramedVectorData::OnCalcStatistics
// apply it to each series after the frame loop (SampleAnalyser.cpp:1065, 2402-2412).
//
// One wave per (buffer, record column) for series of more than 128 frames.  A series of n <= 1024 frames sits in registers as
// x[16 lane' + i] ... (position p = R lane + reg, R = 1, 2, 4, 8 or 16 registers per lane picked
// from n), the moments are wave reductions, the median comes from a bitonic sort of
// order-preserving 64-bit keys (Statistics.cpp:316-413 selects element (n-1)/2 of the sorted
// series, i.e. the lower median).  Results: double[n_bufs][stride][13].

#include <hip/hip_runtime.h>

#include "afx_internal.h"
#include "afx_device.h"

namespace afx {
namespace {

using u64 = unsigned long long;

__device__ __forceinline__ u64 shfl_xor64(u64 v, int m) {
  const int lo = __shfl_xor((int)(unsigned)v, m);
  const int hi = __shfl_xor((int)(unsigned)(v >> 32), m);
  return ((u64)(unsigned)hi << 32) | (unsigned)lo;
}

// bitonic sort of R x 64 keys, p = R lane + reg, all comparators ascending ("flip" form)
template <int R, int J>
__device__ __forceinline__ void stat_inlane(u64 (&key)[R]) {
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int l = i ^ J;
    if (l > i && l < R) {
      const u64 a = key[i], b = key[l];
      key[i] = a < b ? a : b;
      key[l] = a < b ? b : a;
    }
  }
}
template <int R, int M, bool FLIP>
__device__ __forceinline__ void stat_crosslane(u64 (&key)[R], int lane) {
  constexpr int TOP = ((M & (M + 1)) == 0) ? (M + 1) >> 1 : M;
  const bool lower = (lane & TOP) == 0;
  u64 out[R];
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const u64 a = key[i];
    const u64 p = shfl_xor64(key[FLIP ? (R - 1 - i) : i], M);
    const u64 lo = a < p ? a : p, hi = a < p ? p : a;
    out[i] = lower ? lo : hi;
  }
#pragma unroll
  for (int i = 0; i < R; ++i) key[i] = out[i];
}
template <int R, int J>
__device__ __forceinline__ void stat_tail(u64 (&key)[R], int lane) {
  if constexpr (J >= R) stat_crosslane<R, J / R, false>(key, lane);
  else stat_inlane<R, J>(key);
  if constexpr (J > 1) stat_tail<R, J / 2>(key, lane);
}
template <int R, int K>
__device__ __forceinline__ void stat_sort(u64 (&key)[R], int lane) {
  if constexpr (K > 2) stat_sort<R, K / 2>(key, lane);
  if constexpr (K <= R) stat_inlane<R, K - 1>(key);
  else stat_crosslane<R, K / R - 1, true>(key, lane);
  if constexpr (K >= 4) stat_tail<R, K / 4>(key, lane);
}

__device__ __forceinline__ u64 order_key(double v) {
  const u64 b = (u64)__double_as_longlong(v);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_value(u64 k) {
  const u64 b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
  return __longlong_as_double((long long)b);
}

// statistics of one series held as x[reg] = series[R lane + reg]; n >= 2
template <int R>
__device__ void series_stats(const double* col, int64_t cstride, int n, int lane, double* out) {
  double x[R], nxt[R];
  bool ok[R];
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int p = R * lane + i;
    ok[i] = p < n;
    x[i] = ok[i] ? col[(int64_t)p * cstride] : 0.0;
    nxt[i] = (p + 1 < n) ? col[(int64_t)(p + 1) * cstride] : 0.0;
  }
  const double dn = (double)n;
  double mn = 1.0e308, mx = -1.0e308, s = 0.0, sj = 0.0, slog = 0.0, sd = 0.0;
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int p = R * lane + i;
    mn = ok[i] ? fmin(mn, x[i]) : mn;
    mx = ok[i] ? fmax(mx, x[i]) : mx;
    s += x[i];
    sj += (double)p * x[i];
    slog += ok[i] ? fast_log(fabs(x[i]) + 1e-20) : 0.0;    // GeometricMean, Statistics.cpp:417-455
    sd += (p + 1 < n) ? fabs(nxt[i] - x[i]) : 0.0;
  }
  mn = wave_min(mn);
  mx = wave_max(mx);
  s = wave_sum(s);
  sj = wave_sum(sj);
  slog = wave_sum(slog);
  sd = wave_sum(sd);
  const double mean = s / dn;                                // Mean, n >= 2
  const double gmean = fast_exp(slog / dn);
  const double cen = (s == 0.0) ? 0.0 : sj / s;              // Centroid, Statistics.cpp:459-477
  const int nd = n - 1;
  const double dmean = (nd >= 2) ? sd / (double)nd : sd;     // Mean of the n-1 absolute differences
  double var = 0.0, sv = 0.0, dvar = 0.0;
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int p = R * lane + i;
    const double t = x[i] - mean;
    var += ok[i] ? t * t : 0.0;
    const double u = (double)p - cen;
    sv += ok[i] ? u * u * x[i] : 0.0;
    const double w = fabs(nxt[i] - x[i]) - dmean;
    dvar += (p + 1 < n) ? w * w : 0.0;
  }
  var = wave_sum(var) / dn;
  sv = wave_sum(sv);
  dvar = wave_sum(dvar);
  const double spr = (s == 0.0) ? 0.0 : sv / s;              // Spread, Statistics.cpp:486-506
  double sk = 0.0, ku = 0.0;
  if (fabs(spr) > (double)1e-12f) {                          // Statistics.cpp:510-554
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const double t = (x[i] - cen) / spr;
      const double tt = t * t;
      sk += ok[i] ? tt * t : 0.0;
      ku += ok[i] ? tt * tt : 0.0;
    }
    sk = wave_sum(sk) / dn;
    ku = wave_sum(ku) / dn - 3.0;
  }
  // median: element (n-1)/2 of the sorted series
  u64 key[R];
#pragma unroll
  for (int i = 0; i < R; ++i) key[i] = ok[i] ? order_key(x[i]) : ~0ull;
  stat_sort<R, 64 * R>(key, lane);
  const int pm = (n - 1) / 2;
  u64 km = 0;
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int lo = __shfl((int)(unsigned)key[i], pm / R), hi = __shfl((int)(unsigned)(key[i] >> 32), pm / R);
    if ((pm % R) == i) km = ((u64)(unsigned)hi << 32) | (unsigned)lo;
  }
  if (lane == 0) {
    out[0] = mn; out[1] = mx; out[2] = key_value(km); out[3] = mean; out[4] = gmean; out[5] = var;
    out[6] = cen; out[7] = spr; out[8] = sk; out[9] = ku;
    out[10] = (mean == 0.0) ? 0.0 : gmean / mean;            // Flatness, Statistics.cpp:565-574
    out[11] = (n > 2) ? dmean : 0.0;
    out[12] = (n > 2) ? ((nd >= 2) ? dvar / (double)nd : 0.0) : 0.0;
  }
}

// ---- long series (n > 1024 frames: only with the 20 s cap disabled): one wave streams the column ----
// Three passes over the column for the moments (the same formulas as series_stats) and an exact median by a
// most-significant-digit radix select on the order-preserving keys: eight passes of eight bits, each a 256-bin
// histogram (in this wave's LDS slice) of the keys that match the digits chosen so far.
__device__ void series_stats_long(const double* col, int64_t cstride, int n, int lane, unsigned* hist, double* out) {
  const double dn = (double)n;
  double mn = 1.0e308, mx = -1.0e308, s = 0.0, sj = 0.0, slog = 0.0, sd = 0.0;
  for (int p = lane; p < n; p += 64) {
    const double x = col[(int64_t)p * cstride];
    mn = fmin(mn, x);
    mx = fmax(mx, x);
    s += x;
    sj += (double)p * x;
    slog += fast_log(fabs(x) + 1e-20);
    if (p + 1 < n) sd += fabs(col[(int64_t)(p + 1) * cstride] - x);
  }
  mn = wave_min(mn);
  mx = wave_max(mx);
  s = wave_sum(s);
  sj = wave_sum(sj);
  slog = wave_sum(slog);
  sd = wave_sum(sd);
  const double mean = s / dn;
  const double gmean = fast_exp(slog / dn);
  const double cen = (s == 0.0) ? 0.0 : sj / s;
  const int nd = n - 1;
  const double dmean = sd / (double)nd;   // nd >= 2 here
  double var = 0.0, sv = 0.0, dvar = 0.0;
  for (int p = lane; p < n; p += 64) {
    const double x = col[(int64_t)p * cstride];
    const double t = x - mean;
    var += t * t;
    const double u = (double)p - cen;
    sv += u * u * x;
    if (p + 1 < n) {
      const double w = fabs(col[(int64_t)(p + 1) * cstride] - x) - dmean;
      dvar += w * w;
    }
  }
  var = wave_sum(var) / dn;
  sv = wave_sum(sv);
  dvar = wave_sum(dvar);
  const double spr = (s == 0.0) ? 0.0 : sv / s;
  double sk = 0.0, ku = 0.0;
  if (fabs(spr) > (double)1e-12f) {
    for (int p = lane; p < n; p += 64) {
      const double t = (col[(int64_t)p * cstride] - cen) / spr;
      const double tt = t * t;
      sk += tt * t;
      ku += tt * tt;
    }
    sk = wave_sum(sk) / dn;
    ku = wave_sum(ku) / dn - 3.0;
  }
  // median: the key of rank (n-1)/2
  u64 prefix = 0;
  int rank = (n - 1) / 2;
  for (int shift = 56; shift >= 0; shift -= 8) {
    wave_lds_fence();
#pragma unroll
    for (int i = 0; i < 4; ++i) hist[4 * lane + i] = 0;
    wave_lds_fence();
    for (int p = lane; p < n; p += 64) {
      const u64 k = order_key(col[(int64_t)p * cstride]);
      const bool match = (shift == 56) || ((k >> (shift + 8)) == (prefix >> (shift + 8)));
      if (match) atomicAdd(&hist[(unsigned)(k >> shift) & 255u], 1u);
    }
    wave_lds_fence();
    unsigned c[4], local = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      c[i] = hist[4 * lane + i];
      local += c[i];
    }
    const unsigned incl = (unsigned)wave_scan_incl((int)local);
    const unsigned base = incl - local;
    const bool mine = (unsigned)rank >= base && (unsigned)rank < incl;
    int digit = 0, below = 0;
    if (mine) {
      unsigned cum = base;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if ((unsigned)rank >= cum + c[i]) cum += c[i];
        else { digit = 4 * lane + i; below = (int)cum; break; }
      }
    }
    const int src = __ffsll((unsigned long long)__ballot(mine)) - 1;
    digit = __shfl(digit, src);
    below = __shfl(below, src);
    prefix |= (u64)(unsigned)digit << shift;
    rank -= below;
  }
  if (lane == 0) {
    out[0] = mn; out[1] = mx; out[2] = key_value(prefix); out[3] = mean; out[4] = gmean; out[5] = var;
    out[6] = cen; out[7] = spr; out[8] = sk; out[9] = ku;
    out[10] = (mean == 0.0) ? 0.0 : gmean / mean;
    out[11] = dmean;
    out[12] = dvar / (double)nd;
  }
}

// ---- short series (2 <= n <= 128 frames): one LANE per series ----
// A wave takes 64 consecutive series of the batch -- series s = buffer * stride + record column, so the lanes of a wave
// may belong to two buffers with different frame counts: every lane has its own n -- and frame p of a buffer's
// columns is one coalesced row.  The moments are plain sequential sums (the reference's own order).  The median of
// series of up to 64 frames comes from a sorting network in the lane's registers (Batcher's odd-even merge sort: 543
// min / max pairs for 64 values, slots behind n hold +inf), of longer ones from two sorted halves and a bisection.
constexpr int kSmallMax = 128;     // longest series this kernel takes (64 KiB of LDS per wave)
constexpr int kNetwork = 64;       // series up to this length are sorted in registers

// Batcher's odd-even merge sort on x[0 .. N), N a power of two, as compile-time recursion: every comparator is one
// v_min_f64 + one v_max_f64 on two registers (ascending)
template <int I, int J>
__device__ __forceinline__ void net_compare(double (&x)[kNetwork]) {
  const double a = x[I], b = x[J];
  x[I] = fmin(a, b);
  x[J] = fmax(a, b);
}
template <int I, int END, int R, int M>
__device__ __forceinline__ void net_merge_row(double (&x)[kNetwork]) {
  if constexpr (I + R < END) {
    net_compare<I, I + R>(x);
    net_merge_row<I + M, END, R, M>(x);
  }
}
template <int LO, int N, int R>
__device__ __forceinline__ void net_merge(double (&x)[kNetwork]) {
  constexpr int M = 2 * R;
  if constexpr (M < N) {
    net_merge<LO, N, M>(x);
    net_merge<LO + R, N, M>(x);
    net_merge_row<LO + R, LO + N, R, M>(x);
  } else {
    net_compare<LO, LO + R>(x);
  }
}
template <int LO, int N>
__device__ __forceinline__ void net_sort(double (&x)[kNetwork]) {
  if constexpr (N > 1) {
    net_sort<LO, N / 2>(x);
    net_sort<LO + N / 2, N / 2>(x);
    net_merge<LO, N, 1>(x);
  }
}
template <int N>
__device__ __forceinline__ double network_median(const double* tile, int n, int target) {
  double x[kNetwork];
#pragma unroll
  for (int i = 0; i < N; ++i) x[i] = (i < n) ? tile[64 * i] : __builtin_huge_val();
  net_sort<0, N>(x);
  double med = x[0];
#pragma unroll
  for (int i = 1; i < N; ++i) med = (target == i) ? x[i] : med;
  return med;
}

// one wave per workgroup; the tile is small_rows x 64 doubles (small_rows = longest short series of the batch),
// so a CU holds as many waves as fit its 160 KiB of LDS
// (LONG: the batch has series of 65..128 frames; the variant without that path keeps 20 registers more for the rest)
template <bool LONG>
__global__ __launch_bounds__(64) void stats_small_kernel(const StatsArgs a) {
  extern __shared__ double tile_raw[];
  const int lane = threadIdx.x;
  double* const tile = tile_raw + lane;
  const int64_t series = (int64_t)a.n_bufs * a.stride;
  const int64_t total = (series + 63) / 64;
  for (int64_t w = blockIdx.x; w < total; w += gridDim.x) {
    const int64_t s = 64 * w + lane;
    const int64_t buf = (s < series) ? s / a.stride : 0;
    const int col = (int)(s - buf * a.stride);
    const int64_t f0 = a.frame_offset[buf];
    const int64_t nn = a.frame_offset[buf + 1] - f0;
    const bool active = s < series && nn >= 2 && nn <= kSmallMax;     // other lengths: stats_kernel
    const int n = active ? (int)nn : 0;
    const int nmax = wave_max_i(n);
    if (nmax == 0) continue;
    const double* const base = a.rec + f0 * a.stride + (active ? col : 0);
    // the tile in LDS: frame p of this lane's series at tile[64 p], zeros behind the lane's own n.  Rows are asked for
    // sixteen at a time: one load per row waited for in turn was most of this kernel's time
    wave_lds_fence();
    for (int p0 = 0; p0 < nmax; p0 += 16) {
      double t[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) t[k] = (p0 + k < n) ? base[(int64_t)(p0 + k) * a.stride] : 0.0;
#pragma unroll
      for (int k = 0; k < 16; ++k)
        if (p0 + k < nmax) tile[64 * (p0 + k)] = t[k];
    }
    wave_lds_fence();
    const double dn = (double)n;
    double mn = tile[0], mx = mn, sum = 0.0, sj = 0.0, slog = 0.0, sd = 0.0;
    {
      double v = tile[0];
#pragma unroll 4
      for (int p = 0; p < nmax; ++p) {
        const bool ok = p < n, has_next = p + 1 < n;
        const double nxt = (p + 1 < nmax) ? tile[64 * (p + 1)] : v;
        mn = ok ? fmin(mn, v) : mn;
        mx = ok ? fmax(mx, v) : mx;
        sum += v;                                          // (zeros behind n)
        sj += (double)p * v;
        slog += ok ? fast_log(fabs(v) + 1e-20) : 0.0;      // GeometricMean, Statistics.cpp:417-455
        sd += has_next ? fabs(nxt - v) : 0.0;
        v = nxt;
      }
    }
    const double mean = sum / dn;
    const double gmean = fast_exp(slog / dn);
    const double cen = (sum == 0.0) ? 0.0 : sj / sum;      // Centroid, Statistics.cpp:459-477
    const int nd = n - 1;
    const double dmean = (nd >= 2) ? sd / (double)nd : sd;
    const double spr_den = sum;
    double var = 0.0, sv = 0.0, dvar = 0.0;
    {
      double v = tile[0];
#pragma unroll 4
      for (int p = 0; p < nmax; ++p) {
        const bool ok = p < n, has_next = p + 1 < n;
        const double nxt = (p + 1 < nmax) ? tile[64 * (p + 1)] : v;
        const double t = v - mean;
        var += ok ? t * t : 0.0;
        const double u = (double)p - cen;
        sv += u * u * v;                                   // (zeros behind n)
        const double wd = fabs(nxt - v) - dmean;
        dvar += has_next ? wd * wd : 0.0;
        v = nxt;
      }
    }
    var /= dn;
    const double spr = (spr_den == 0.0) ? 0.0 : sv / spr_den;   // Spread, Statistics.cpp:486-506
    double sk = 0.0, ku = 0.0;
    {
      // Statistics.cpp:510-554; (x - c) / v as (x - c) (1 / v): <= 1 ulp apart.  Lanes whose spread is below the
      // cut run the loop on a harmless reciprocal and drop the result.
      const bool on = fabs(spr) > (double)1e-12f;
      const double rspr = on ? 1.0 / spr : 0.0;
#pragma unroll 4
      for (int p = 0; p < nmax; ++p) {
        const double t = (tile[64 * p] - cen) * rspr;
        const double tt = t * t;
        sk += (p < n) ? tt * t : 0.0;
        ku += (p < n) ? tt * tt : 0.0;
      }
      sk = on ? sk / dn : 0.0;
      ku = on ? ku / dn - 3.0 : 0.0;
    }
    // median: element (n-1)/2 of the sorted series (Statistics.cpp:316-413 selects the lower median)
    const int target = (n - 1) / 2;
    double med = mn;
    if (nmax <= 16) med = network_median<16>(tile, n, target);
    else if (nmax <= 32) med = network_median<32>(tile, n, target);
    else if (!LONG || nmax <= kNetwork) med = network_median<kNetwork>(tile, n, target);
    else {
      // 65..128 frames: the two halves of the series are sorted by the same network, one after the other, and written
      // back over the tile (nothing reads it in frame order any more); the element of rank `target` of their union is
      // then found by bisection on how many elements the first half contributes (the classic selection from two sorted
      // arrays): max(A[i - 1], B[target - i]) for the i with A[i - 1] <= B[target + 1 - i] and B[target - i] <= A[i]
      {
        double x[kNetwork];
#pragma unroll
        for (int i = 0; i < kNetwork; ++i) x[i] = (i < n) ? tile[64 * i] : __builtin_huge_val();
        net_sort<0, kNetwork>(x);
        wave_lds_fence();
#pragma unroll
        for (int i = 0; i < kNetwork; ++i) tile[64 * i] = x[i];
#pragma unroll
        for (int i = 0; i < kNetwork; ++i) x[i] = (kNetwork + i < n) ? tile[64 * (kNetwork + i)] : __builtin_huge_val();
        net_sort<0, kNetwork>(x);
#pragma unroll
        for (int i = 0; i < kNetwork; ++i)
          if (kNetwork + i < nmax) tile[64 * (kNetwork + i)] = x[i];
        wave_lds_fence();
      }
      const int na = min(n, kNetwork), nb = max(n - kNetwork, 0);      // lengths of the sorted halves A, B
      // i = elements of A among the target + 1 smallest: in [max(0, target + 1 - nb), min(target + 1, na)]
      int lo = max(0, target + 1 - nb), hi = min(target + 1, na);
      const double inf = __builtin_huge_val();
      auto A = [&](int i) { return (i < 0) ? -inf : ((i >= na) ? inf : tile[64 * i]); };
      auto B = [&](int j) { return (j < 0) ? -inf : ((j >= nb) ? inf : tile[64 * (kNetwork + j)]); };
#pragma unroll 1
      for (int it = 0; it < 8; ++it) {          // 2^7 >= 65 candidates; lanes that are done keep lo == hi
        const int i = (lo + hi) >> 1, j = target + 1 - i;
        // too few from A when B[j - 1] > A[i]
        const bool more = lo < hi && B(j - 1) > A(i);
        lo = more ? i + 1 : lo;
        hi = more ? hi : i;
      }
      med = fmax(A(lo - 1), B(target - lo));
    }
    if (active) {
      double* const out = a.stats + s * 13;
      out[0] = mn; out[1] = mx; out[2] = med; out[3] = mean; out[4] = gmean; out[5] = var;
      out[6] = cen; out[7] = spr; out[8] = sk; out[9] = ku;
      out[10] = (mean == 0.0) ? 0.0 : gmean / mean;        // Flatness, Statistics.cpp:565-574
      out[11] = (n > 2) ? dmean : 0.0;
      out[12] = (n > 2) ? ((nd >= 2) ? dvar / (double)nd : 0.0) : 0.0;
    }
  }
}

__global__ __launch_bounds__(256) void stats_kernel(const StatsArgs a) {
  __shared__ unsigned hist_all[4 * 256];   // radix-select histograms of the long-series path, one slice per wave
  unsigned* const hist = hist_all + 256 * (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t total = (int64_t)a.n_bufs * a.stride;
  for (int64_t w = wave0; w < total; w += (int64_t)gridDim.x * 4) {
    const int64_t buf = w / a.stride;
    const int colidx = (int)(w - buf * a.stride);
    const int64_t f0 = a.frame_offset[buf];
    const int64_t nn = a.frame_offset[buf + 1] - f0;
    double* const out = a.stats + w * 13;
    const double* const col = a.rec + f0 * a.stride + colidx;
    if (nn > 1024) {   // does not fit one wave's registers: streamed
      series_stats_long(col, a.stride, (int)nn, lane, hist, out);
      continue;
    }
    const int n = (int)nn;
    if (n >= 2 && n <= kSmallMax) continue;   // stats_small_kernel
    if (n <= 1) {      // Statistics.cpp:72-89: only min/max/mean (and zeros) are assigned
      if (lane < 13) out[lane] = 0.0;
      if (n == 1 && lane == 0) {
        const double v = col[0];
        out[0] = v; out[1] = v; out[3] = v;
      }
      continue;
    }
    if (n <= 64) series_stats<1>(col, a.stride, n, lane, out);
    else if (n <= 128) series_stats<2>(col, a.stride, n, lane, out);
    else if (n <= 256) series_stats<4>(col, a.stride, n, lane, out);
    else if (n <= 512) series_stats<8>(col, a.stride, n, lane, out);
    else series_stats<16>(col, a.stride, n, lane, out);
  }
}

}  // namespace

hipError_t launch_stats(const StatsArgs& a, hipStream_t stream) {
  const int64_t total = (int64_t)a.n_bufs * a.stride;
  if (total <= 0) return hipSuccess;
  const int64_t want = (total + 3) / 4;
  const int grid = (int)(want < 256 * 32 ? want : 256 * 32);
  hipError_t e = hipSuccess;
  if (a.need_long) {
    hipLaunchKernelGGL(stats_kernel, dim3(grid), dim3(256), 0, stream, a);
    if ((e = hipGetLastError()) != hipSuccess) return e;
  }
  // short series: lane-per-series kernel; 32 KiB of LDS per wave
  static bool raised[16] = {};
  if (a.small_rows <= 0) return hipSuccess;        // no buffer with 2..128 frames in this batch
  const int lds = a.small_rows * 64 * 8;
  constexpr int lds_max = kSmallMax * 64 * 8;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 16 || !raised[dev]) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(stats_small_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(stats_small_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 16) raised[dev] = true;
  }
  const int64_t waves = ((int64_t)a.n_bufs * a.stride + 63) / 64;
  if (a.small_rows > kNetwork) hipLaunchKernelGGL(stats_small_kernel<true>, dim3((unsigned)(waves < 16384 ? waves : 16384)), dim3(64), lds, stream, a);
  else hipLaunchKernelGGL(stats_small_kernel<false>, dim3((unsigned)(waves < 16384 ? waves : 16384)), dim3(64), lds, stream, a);
  return hipGetLastError();
}

}  // namespace afx
