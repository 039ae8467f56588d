// afec_amd/csrc/afx_bands.hip -- the 14 "sub-band" descriptors of
// TSampleAnalyser::CalcSpectralBandFeatures (SampleAnalyser.cpp:2067-2308) on gfx950.
//
// One wave per chunk of consecutive frames (the frame kernel's chunk table), working from the magnitude
// spectra the frame kernel left in HBM ([F][1024] doubles, L2/MALL-resident between the two launches for
// realistic batch sizes).  The previous frame's spectrum and band sums (flux) are carried in registers
// from frame to frame; only a chunk's first frame reads its predecessor.  Bands are laid out contiguously from
// bin 1 with the reference's bin counts (the drift from the nominal edges is intentional, see
// SURVEY 8a/a15), so every membership test below is a compile-time range of k = 64 r + lane.
//
//   rms, flatness(dB), flux      masked per-band sums, reduced 16 bands at a time (wave_sum16)
//   complexity                   band maximum (wave_max16) -> threshold -> strict local maxima
//   contrast                     "mean of the n lowest / highest bins of the band" (std::sort + two loops,
//                                SA:2200-2228) as an exact selection: bitonic sorts of blocks of 256 positions, in
//                                registers, on the 32-bit key (band << 27 | float bits >> 4) only find, per band, the
//                                key at the cut; the sums are then taken over the ORIGINAL doubles of the bins
//                                whose key lies on the right side of the cut.  Bins whose keys tie with the
//                                cut's (values equal to 19 mantissa bits) are all on one side in the common
//                                case; when the cut goes through such a tie class it is resolved exactly on
//                                the doubles (exact_cut_sum)

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
// spectral_flux is the sub-bands' flux formula over bins 1..738: its sums ride along as band 14 of the per-band results
constexpr int kWholeBand = 14;
constexpr int kParked = 9;   // per-band sums of a frame that wait for the group's closed forms

__attribute__((always_inline)) constexpr bool sub_touches(int b, int r) { return kSubStart[b] <= 64 * r + 63 && kSubStart[b + 1] - 1 >= 64 * r; }
// position range of band b in the sorted array (bands are sorted by id first)
__attribute__((always_inline)) constexpr int sub_pos0(int b) { return kSubStart[b] - 1; }
__attribute__((always_inline)) constexpr bool pos_touches(int lo, int hi, int lane_lo) { return lo <= lane_lo + 15 && hi - 1 >= lane_lo; }

__device__ __forceinline__ bool in_band(int b, int r, int lane) {
  const int k = 64 * r + lane;
  return k >= kSubStart[b] && k < kSubStart[b + 1];
}
// does band b cover every bin of row r?
__attribute__((always_inline)) constexpr bool sub_covers(int b, int r) { return kSubStart[b] <= 64 * r && kSubStart[b + 1] - 1 >= 64 * r + 63; }
// (band, row) pairs that touch, in band-major order
__attribute__((always_inline)) constexpr int sub_pair_count() {
  int n = 0;
  for (int b = 0; b < kNumSub; ++b)
    for (int r = 0; r < kRows; ++r) n += sub_touches(b, r) ? 1 : 0;
  return n;
}
static_assert(sub_pair_count() == 25, "(band, row) pairs of the 64-lane layout");
__attribute__((always_inline)) constexpr int sub_pair_index(int b, int r) {
  int n = 0;
  for (int bb = 0; bb < kNumSub; ++bb)
    for (int rr = 0; rr < kRows; ++rr) {
      if (bb == b && rr == r) return n;
      n += sub_touches(bb, rr) ? 1 : 0;
    }
  return -1;
}

// v in the lanes whose bit is set in the wave-uniform mask m, 0.0 elsewhere: two v_cndmask_b32 reading the
// mask from an SGPR pair.  The band-membership masks depend on the lane only, so they are built once per
// wave (ballots) instead of two compares per (band, row) pair, quantity and frame.
using mask64 = unsigned long long;
__device__ __forceinline__ double keep_where(double v, mask64 m) {
  int lo, hi;
  asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(lo) : "v"(__double2loint(v)), "s"(m));
  asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(hi) : "v"(__double2hiint(v)), "s"(m));
  return __hiloint2double(hi, lo);
}

// v where the mask's bit is set, 1.0 elsewhere (factors of a product)
__device__ __forceinline__ double keep_or_one(double v, mask64 m) {
  int lo, hi;
  asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(lo) : "v"(__double2loint(v)), "s"(m));
  asm("v_cndmask_b32_e64 %0, %3, %1, %2" : "=v"(hi) : "v"(__double2hiint(v)), "s"(m), "v"(0x3FF00000));
  return __hiloint2double(hi, lo);
}

// Lane masks of the (band, row) pairs as compile-time constants: which lanes of row r (bins 64 r .. 64 r + 63) hold bins
// [lo, hi).  They depend on nothing but the lane, so they are 64-bit literals the scalar unit materialises where they are
// used (two s_mov_b32, off the vector pipe) -- until round 5 the sub-bands' masks were 25 ballots held in 50 SGPRs for the
// whole kernel and the spectrum bands' were two v_cmp per pair and frame.
__attribute__((always_inline)) constexpr mask64 row_mask(int r, int lo, int hi) {
  const int l0 = lo - 64 * r < 0 ? 0 : (lo - 64 * r > 64 ? 64 : lo - 64 * r);   // lanes [l0, l1)
  const int l1 = hi - 64 * r < 0 ? 0 : (hi - 64 * r > 64 ? 64 : hi - 64 * r);
  const mask64 below_l1 = l1 >= 64 ? ~0ull : ((1ull << l1) - 1ull);
  const mask64 below_l0 = l0 >= 64 ? ~0ull : ((1ull << l0) - 1ull);
  return l1 > l0 ? (below_l1 & ~below_l0) : 0ull;
}
// (as tables: an unrolled loop's index into a constant table folds to the literal; the expression itself did not)
struct SubMaskTable { mask64 m[kNumSub][kRows]; };
struct SpectrumMaskTable { mask64 m[kNumBands][kRows]; };
constexpr SubMaskTable make_sub_masks() {
  SubMaskTable t{};
  for (int b = 0; b < kNumSub; ++b)
    for (int r = 0; r < kRows; ++r) t.m[b][r] = row_mask(r, kSubStart[b], kSubStart[b + 1]);
  return t;
}
constexpr SpectrumMaskTable make_spectrum_masks() {
  SpectrumMaskTable t{};
  for (int b = 0; b < kNumBands; ++b)
    for (int r = 0; r < kRows; ++r) t.m[b][r] = row_mask(r, kBandEdge[b], kBandEdge[b + 1]);
  return t;
}
constexpr SubMaskTable kSubMasks = make_sub_masks();
constexpr SpectrumMaskTable kSpectrumMasks = make_spectrum_masks();
__attribute__((always_inline)) constexpr mask64 sub_mask(int b, int r) { return kSubMasks.m[b][r]; }
__attribute__((always_inline)) constexpr mask64 spectrum_mask(int b, int r) { return kSpectrumMasks.m[b][r]; }
constexpr mask64 kFirstRowOk = row_mask(0, kFirstBin, 64);                               // bins 1..63
constexpr mask64 kLastRowOk = row_mask(kRows - 1, 64 * (kRows - 1), kLastBin + 1);       // bins 704..738

// per-band sum of one value per row; lane L receives the total of band (L >> 2) & 15
// whole: one more per-lane value, summed over the wave into "band" kWholeBand
template <typename F>
__device__ __forceinline__ double band_sum(F value_of_row, int lane, double whole = 0.0) {
  double acc[16];
#pragma unroll
  for (int b = 0; b < 16; ++b) {
    acc[b] = (b == kWholeBand) ? whole : 0.0;
    if (b < kNumSub) {
      bool first = true;   // (0.0 + v is not v for the compiler: v may be -0.0)
#pragma unroll
      for (int r = 0; r < kRows; ++r)
        if (sub_touches(b, r)) {
          const double v = sub_covers(b, r) ? value_of_row(r) : keep_where(value_of_row(r), sub_mask(b, r));
          acc[b] = first ? v : acc[b] + v;
          first = false;
        }
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
      bool first = true;   // the values are magnitudes: >= 0
#pragma unroll
      for (int r = 0; r < kRows; ++r)
        if (sub_touches(b, r)) {
          const double v = sub_covers(b, r) ? value_of_row(r) : keep_where(value_of_row(r), sub_mask(b, r));
          acc[b] = first ? v : fmax(acc[b], v);
          first = false;
        }
    }
  }
  return wave_max16(acc, lane);
}

// The spectrum bands ("frequency_bands", SA:2007-2048) that lie inside the stored rows: bands 0..25 = bins 1..737 (the
// half-wave frame kernel sums bands 26, 27 = bins 738..1023 itself, from the mirrored halves it does not store).
constexpr int kBandsHere = 26;
// xx[r] = |X[64 r + lane]|^2; lanes with (lane & 3) == h, h = 0 / 1, receive band 16 h + (lane >> 2).
__device__ __forceinline__ double spectrum_band_sums(const double (&xx)[12], int lane) {
  double mine = 0.0;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    double acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int b = 16 * half + i;
      acc[i] = 0.0;
      if (b < kBandsHere) {
        bool first = true;
#pragma unroll
        for (int r = 0; r < 12; ++r)
          if (band_touches(b, r)) {
            const bool whole = kBandEdge[b] <= 64 * r && kBandEdge[b + 1] - 1 >= 64 * r + 63;
            const double v = whole ? xx[r] : keep_where(xx[r], spectrum_mask(b, r));
            acc[i] = first ? v : acc[i] + v;
            first = false;
          }
      }
    }
    const double tot = wave_sum16(acc, lane);  // lane L: band 16 half + ((L >> 2) & 15)
    if ((lane & 3) == half) mine = tot;
  }
  return mine;
}

// sqrt(denom2) of TStatistics::Correlation (Statistics.cpp:631-635): a product of two sums of squares that rounding
// made negative gives NaN there, and here
__device__ __forceinline__ double pearson_root(double denom2) {
  return denom2 > 0.0 ? mag_sqrt(denom2) : __builtin_nan("");
}

// natural log for the geometric means of the bands (Statistics.cpp:417-455): the sum of up to 287 of these is divided by
// the band's bin count and exponentiated, so 1e-13 absolute on one logarithm is far below what the descriptor resolves.
// x = 2^e m, m in [0.5, 1); the top six mantissa bits pick an interval with centre c: log m = log1p(m / c - 1) + log c,
// with 1 / c (rounded) and -log(1 / c) from a 64-entry table in LDS, |m / c - 1| < 2^-7 and the series cut after r^6 / 6
// (the next term is 3e-16).  15 instructions and one LDS read against the 30 of the atanh form with its quotient.
struct LogEntry { double inv_c, log_c; };
__device__ __forceinline__ void band_log_table(LogEntry* table, int lane) {
  const double inv_c = 1.0 / (0.5 + ((double)lane + 0.5) * (1.0 / 128.0));
  table[lane] = {inv_c, -fast_log(inv_c)};
}
__device__ __forceinline__ double band_log(double x, const LogEntry* table) {
  const double m = __builtin_amdgcn_frexp_mant(x);   // [0.5, 1)
  const int e = __builtin_amdgcn_frexp_exp(x);
  const LogEntry t = table[((unsigned)__double2hiint(m) >> 14) & 63u];
  const double r = fma(m, t.inv_c, -1.0);
  double q = fma(r, -1.0 / 6.0, 1.0 / 5.0);
  q = fma(q, r, -1.0 / 4.0);
  q = fma(q, r, 1.0 / 3.0);
  q = fma(q, r, -1.0 / 2.0);
  return fma((double)e, 0.693147180559945309417, t.log_c) + fma(q, r * r, r);
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
  // the lane of a pair with the TOP bit clear keeps the smaller key, the other the larger: median of (own, partner, 0) and
  // of (own, partner, ~0) -- one v_med3_u32 per key instead of minimum, maximum and a select
  const u32 lim = (u32)__builtin_amdgcn_sbfe(lane, __builtin_ctz(TOP), 1);
  u32 out[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const u32 p = lane_xor_u32<M>(key[i ^ RX]);
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(out[i]) : "v"(key[i]), "v"(p), "v"(lim));
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
// a lane's sixteen keys ascending: the 60-comparator network for 16 inputs (ten layers; checked on all 2^16 0-1 inputs) --
// the levels K <= 16 of the bitonic network are 80 comparators
__device__ __forceinline__ void sort16_inlane(u32 (&key)[16]) {
  constexpr int kNet[60][2] = {
      {0, 13}, {1, 12}, {2, 15}, {3, 14}, {4, 8}, {5, 6}, {7, 11}, {9, 10}, {0, 5}, {1, 7}, {2, 9}, {3, 4}, {6, 13}, {8, 14}, {10, 15},
      {11, 12}, {0, 1}, {2, 3}, {4, 5}, {6, 8}, {7, 9}, {10, 11}, {12, 13}, {14, 15}, {0, 2}, {1, 3}, {4, 10}, {5, 11}, {6, 7}, {8, 9},
      {12, 14}, {13, 15}, {1, 2}, {3, 12}, {4, 6}, {5, 7}, {8, 10}, {9, 11}, {13, 14}, {1, 4}, {2, 6}, {5, 8}, {7, 10}, {9, 13}, {11, 14},
      {2, 4}, {3, 6}, {9, 12}, {11, 13}, {3, 5}, {6, 8}, {7, 9}, {10, 12}, {3, 4}, {5, 6}, {7, 8}, {9, 10}, {11, 12}, {6, 7}, {8, 9}};
#pragma unroll
  for (int c = 0; c < 60; ++c) {
    const u32 a = key[kNet[c][0]], b = key[kNet[c][1]];
    key[kNet[c][0]] = a < b ? a : b;
    key[kNet[c][1]] = a < b ? b : a;
  }
}
template <int K>
__device__ __forceinline__ void sort_level(u32 (&key)[16], int lane) {
  if constexpr (K == 16) {
    sort16_inlane(key);
  } else {
    sort_level<K / 2>(key, lane);
    crosslane_stage<(K >> 4) - 1, 15>(key, lane);        // flip across lanes: lane ^ (K/16 - 1), reg ^ 15
    merge_tail<K, K / 4>(key, lane);
  }
}

// sort key of a magnitude inside band b: band << 27 | float bits >> 4 (rounded): 19 mantissa bits order the values
__device__ __forceinline__ u32 sort_key(int band, double v) {
  return ((u32)band << 27) | ((__float_as_uint((float)fabs(v)) + 8u) >> 4);
}

// Sum of the take smallest (valley) or largest (peak) magnitudes of band b when the cut falls inside a class of
// bins with equal sort keys: the bins strictly beyond the cut key are summed as they are, the tie class is
// resolved on the doubles themselves (all equal -> a multiple of the value; else smallest / largest first, a value
// at a time).  Wave-uniform b; x: this lane's magnitudes.  Rare (and cheap for silence: every bin is 0).
template <bool VALLEY>
__device__ __noinline__ double exact_cut_sum(const double* cur, int b, u32 cut_key, int lane) {
  // b is wave-uniform: select chains on the scalar unit instead of run-time indexed tables (which would live in
  // scratch).  The band's bins are re-read from the spectrum (cache-resident) in rolled loops: this path must not
  // cost the main loop registers.
  int start = 0, end = 0, take = 1;
#pragma unroll
  for (int i = 0; i < kNumSub; ++i) {
    start = (b == i) ? kSubStart[i] : start;
    end = (b == i) ? kSubStart[i + 1] : end;
    take = (b == i) ? kSubNeigh[i] : take;
  }
  const int r0 = start >> 6, r1 = (end - 1) >> 6;
  double strict = 0.0, tlo = __builtin_huge_val(), thi = -__builtin_huge_val();
  int n_strict = 0;
#pragma unroll 1
  for (int r = r0; r <= r1; ++r) {
    const int k = 64 * r + lane;
    const bool inb = k >= start && k < end;
    const double xv = cur[k];
    const u32 kx = sort_key(b, xv);
    const bool beyond = inb && (VALLEY ? kx < cut_key : kx > cut_key);
    const bool tie = inb && kx == cut_key;
    strict += beyond ? xv : 0.0;
    n_strict += __popcll(__ballot(beyond));
    tlo = tie ? fmin(tlo, xv) : tlo;
    thi = tie ? fmax(thi, xv) : thi;
  }
  strict = wave_sum(strict);
  tlo = wave_min(tlo);
  thi = wave_max(thi);
  int need = take - n_strict;          // >= 1
  if (tlo == thi) return strict + (double)need * tlo;
  // distinct values inside the tie class: take them in order; everything up to `last` is already taken
  double tsum = 0.0, last = VALLEY ? -__builtin_huge_val() : __builtin_huge_val();
  while (need > 0) {                   // wave-uniform
    double v = VALLEY ? __builtin_huge_val() : -__builtin_huge_val();
#pragma unroll 1
    for (int r = r0; r <= r1; ++r) {
      const int k = 64 * r + lane;
      const double xv = cur[k];
      const bool left = k >= start && k < end && sort_key(b, xv) == cut_key && (VALLEY ? xv > last : xv < last);
      v = left ? (VALLEY ? fmin(v, xv) : fmax(v, xv)) : v;
    }
    v = VALLEY ? wave_min(v) : wave_max(v);
    int c = 0;
#pragma unroll 1
    for (int r = r0; r <= r1; ++r) {
      const int k = 64 * r + lane;
      const double xv = cur[k];
      c += __popcll(__ballot(k >= start && k < end && xv == v && sort_key(b, xv) == cut_key));
    }
    const int t = c < need ? c : need;
    tsum += (double)t * v;
    need -= t;
    last = v;
  }
  return strict + tsum;
}

// LDS-DMA of 16 bytes per lane (lane L's bytes land at lds_base + IMM + 16 L, read from gaddr + IMM): how a frame's
// spectrum reaches the wave's LDS image while the wave works on the frame before.  Inline assembly for the reason
// given in afx_frames32.hip (the builtin makes the compiler put s_waitcnt vmcnt(0) in front of every later DS read);
// the kernel orders it against its own DS traffic by hand.  M0 has no other user here.
template <int IMM>
__device__ __forceinline__ void dma_16(const void* gaddr, unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off offset:%2"
               :: "v"(gaddr), "s"(lds_base), "n"(IMM) : "memory", "m0");
}
// bins 0..767 of one frame (6 KiB): lane L moves bins 2 L, 2 L + 1 of every 128-bin piece
__device__ __forceinline__ void dma_frame(const double* frame, int lane, unsigned lds_image) {
  const unsigned char* const g = reinterpret_cast<const unsigned char*>(frame) + 16 * lane;
  dma_16<0>(g, lds_image);
  dma_16<1024>(g, lds_image);
  dma_16<2048>(g, lds_image);
  dma_16<3072>(g, lds_image);
  dma_16<0>(g + 4096, lds_image + 4096);
  dma_16<1024>(g + 4096, lds_image + 4096);
}

// FLAGS: the kBands* bits as a compile-time constant for the combinations the planner produces for whole descriptor
// sets (what is not selected costs neither instructions nor registers), -1: read BandArgs::flags
template <int FLAGS>
__global__ __launch_bounds__(256, 2) void bands_kernel(const BandArgs a) {
  const int flags = FLAGS >= 0 ? FLAGS : a.flags;
  const int lane = threadIdx.x & 63;
  const int wave0 = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int stride = gridDim.x * 4;
  __shared__ double s_thr[4][16];
  __shared__ u32 s_sorted[4][1024];
  __shared__ u32 s_cut[4][32];
  __shared__ double s_cn[4][48];   // per band: 1 / bins, 1 / neighbours, bins
  __shared__ u32 s_sel[4][8];      // the cut keys of the two bands that lie across a block boundary of the sort
  __shared__ int s_cutpos[4][32];  // per band: position of the valley's cut key, of the peak's
  // The wave's image of the frame it works on, bins 0..767 in natural order: filled by LDS-DMA while the frame before is
  // being worked on (issued once the last read of the image is done, about a third into a frame), so a frame's spectrum
  // is not waited for at the top of the loop; the rolloff walk (kBandsStats) and the local maxima's neighbour bins read
  // it too (until round 5: twelve global loads at the top of every frame, waited for at once; a natural-order copy written
  // back to LDS for the rolloff; 24 more global loads for the neighbours)
  __shared__ __align__(16) double s_nat[4][64 * kRows];
  __shared__ LogEntry s_log[4][64];
  __shared__ double s_park[4][4 * kParked * 16];   // [frame of the group][quantity][band]
  double* const thr = s_thr[threadIdx.x >> 6];
  u32* const sorted = s_sorted[threadIdx.x >> 6];
  u32* const cut = s_cut[threadIdx.x >> 6];      // [0..15] valley cut key per band, [16..31] peak cut key
  double* const cn = s_cn[threadIdx.x >> 6];
  u32* const sel = s_sel[threadIdx.x >> 6];
  int* const cutpos = s_cutpos[threadIdx.x >> 6];
  double* const nat = s_nat[threadIdx.x >> 6];
  typedef __attribute__((address_space(3))) void lvoid;
  const unsigned nat_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lvoid*)nat);   // its LDS byte address (M0 of the DMA)
  using lds_vdouble = volatile __attribute__((address_space(3))) double;
  lds_vdouble* const image = (lds_vdouble*)nat;
  LogEntry* const logt = s_log[threadIdx.x >> 6];
  double* const park = s_park[threadIdx.x >> 6];
  band_log_table(logt, lane);
  if (lane < 16) {
    double nb = 1.0, nn = 1.0;
#pragma unroll
    for (int i = 0; i < kNumSub; ++i) {
      nb = (lane == i) ? (double)kSubN[i] : nb;
      nn = (lane == i) ? (double)kSubNeigh[i] : nn;
    }
    nb = (lane == kWholeBand) ? (double)kBinCount : nb;   // the whole analysis range as one more band (spectral_flux)
    cn[lane] = 1.0 / nb; cn[16 + lane] = 1.0 / nn; cn[32 + lane] = nb;
    int pv = 0, pp = 1;
#pragma unroll
    for (int i = 0; i < kNumSub; ++i) {
      const int pos0 = (i == 12) ? kSubStart[12] : sub_pos0(i);
      pv = (lane == i) ? pos0 + kSubNeigh[i] - 1 : pv;
      pp = (lane == i) ? pos0 + kSubN[i] - kSubNeigh[i] : pp;
    }
    cutpos[lane] = pv; cutpos[16 + lane] = pp;
  }
  wave_lds_fence();

  // lane-only facts, once per wave: membership masks of the (band, row) pairs, the band of each of the
  // lane's bins, the analysis-range masks of the first and last row
  int bid[kRows];
#pragma unroll
  for (int r = 0; r < kRows; ++r) {
    bid[r] = 15;
#pragma unroll
    for (int b = 0; b < kNumSub; ++b)
      if (sub_touches(b, r)) bid[r] = in_band(b, r, lane) ? b : bid[r];
  }

  for (int ci = wave0; ci < a.n_chunks; ci = next_item(a.queue, ci, min(stride, a.n_chunks), stride, lane)) {
    const Chunk ch = a.chunks[ci];
    double x[kRows], y[kRows];
    // the frame before the chunk; the first frame of a buffer is compared with itself (SA:937-940)
    const int first_back = (ch.flags & kChunkFirstOfBuffer) ? 0 : 1;
    if (flags & (kBandsFeatures | kBandsFlux)) {
      const double* const prv = a.mag + ((int64_t)ch.frame0 - first_back) * kHalf;
#pragma unroll
      for (int r = 0; r < kRows; ++r) y[r] = prv[64 * r + lane];
    }
    // sums of the previous frame, carried from frame to frame inside the chunk
    // Sums of products are sums of rounded products (mul_rn) here and in the loop: the previous frame's sums reach a
    // frame either from this prologue or carried from the iteration before, and both must be the same bits.
    double fb = 0.0, fbb = 0.0, sy = 0.0, syy = 0.0;
    if (flags & (kBandsFeatures | kBandsFlux)) {
      double yy[kRows];
#pragma unroll
      for (int r = 0; r < kRows; ++r) yy[r] = mul_rn(y[r], y[r]);
      if (flags & kBandsFlux) {
#pragma unroll
        for (int r = 0; r < kRows; ++r) {
          const bool edge = r == 0 || r == kRows - 1;
          const mask64 ok = (r == 0) ? kFirstRowOk : kLastRowOk;
          fb += edge ? keep_where(y[r], ok) : y[r];
          fbb += edge ? keep_where(yy[r], ok) : yy[r];
        }
      }
      if (flags & kBandsFeatures) {
        // (with the sub-bands selected the sums of spectral_flux over the whole range are band kWholeBand of the reductions)
        sy = band_sum([&](int r) { return y[r]; }, lane, fb);
        syy = band_sum([&](int r) { return yy[r]; }, lane, fbb);
      }
    }
    if (!(flags & kBandsFeatures) && (flags & kBandsFlux)) {
      fb = wave_sum(fb); fbb = wave_sum(fbb);
    }

    // the chunk's first frame (every DS read of the chunk before has been waited for: its results were stored)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    dma_frame(a.mag + (int64_t)ch.frame0 * kHalf, lane, nat_lds);
  for (int fi = 0; fi < ch.nframes; ++fi) {
    // lane-only predicates other than the membership masks (reduction selects, sort directions, position
    // ranges) are recomputed per frame from a re-materialised lane id: hoisted out of the loop they would
    // all sit in SGPR pairs and spill
    int lane_v = lane;
    asm volatile("" : "+v"(lane_v));
    const int64_t f = (int64_t)ch.frame0 + fi;
    const double* const cur = a.mag + f * kHalf;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the frame's image has landed
#pragma unroll
    for (int r = 0; r < kRows; ++r) x[r] = image[64 * r + lane_v];
    // products rounded on their own (mul_rn): see the chunk prologue
    double xx[kRows], xy[kRows];
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
      xx[r] = mul_rn(x[r], x[r]);
      xy[r] = mul_rn(x[r], y[r]);
    }
    // ---- spectrum bands 0..25 for the half-wave frame kernel, which only stores the magnitudes ----
    if (flags & kBandsSpectrum) {
      const double mine = spectrum_band_sums(xx, lane_v);
      const int b = 16 * (lane_v & 3) + (lane_v >> 2);
      if ((lane_v & 3) < 2 && b < kBandsHere) a.rec[f * a.lay.stride + a.lay.bands + b] = mine;
    }
    // ---- the raw sums of the spectral statistics over bins 1..738, j = bin - 1 (SA:1808-1915), for the half-wave
    //      frame kernel's magnitude class: sum m, m^2, j m, j^2 m, m^3, m^4, sum log(m + 1e-20) and the rolloff count, as
    //      its statistics class leaves them for stats32_finish_kernel (the stored magnitudes are flushed already) ----
    if (flags & kBandsStats) {
      double s1 = 0.0, s2 = 0.0, sj = 0.0, sjj = 0.0, s3 = 0.0, s4 = 0.0, prod = 1.0;
      double jq = (double)(lane_v - 1);
#pragma unroll
      for (int r = 0; r < kRows; ++r) {
        const double m = (r == 0) ? keep_where(x[r], kFirstRowOk) : (r == kRows - 1 ? keep_where(x[r], kLastRowOk) : x[r]);
        const double m2 = m * m, jm = jq * m;
        s1 += m;
        s2 += m2;
        sj += jm;
        sjj = fma(jq, jm, sjj);
        s3 = fma(m2, m, s3);
        s4 = fma(m2, m2, s4);
        // (bins outside the range: factor 1)
        const double f = m + 1e-20;
        prod *= (r == 0) ? keep_or_one(f, kFirstRowOk) : (r == kRows - 1 ? keep_or_one(f, kLastRowOk) : f);
        jq += 64.0;
      }
      double st[16];
      st[0] = s1; st[1] = s2; st[2] = sj; st[3] = sjj; st[4] = s3; st[5] = s4; st[6] = fast_log(prod);
#pragma unroll
      for (int i = 7; i < 16; ++i) st[i] = 0.0;
      const double red = wave_sum16(st, lane_v);          // lane L: st[(L >> 2) & 15]
      double* const tmp = a.stat_tmp + f * kStatTmp;
      if ((lane_v & 3) == 0 && (lane_v >> 2) < 7) tmp[lane_v >> 2] = red;
      // rolloff (scalar.c:472-492): bins whose running sum stays below 85 % of the total: natural-order copy in LDS, each
      // lane walks 12 consecutive bins
      const double total = read_lane<0>(red);
      // (the image holds the stored magnitudes as they are: bin 0 is never walked, bins above the range are cut here)
      double seg[12], segsum = 0.0;
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        const int k = kFirstBin + 12 * lane_v + i;
        seg[i] = (k <= kLastBin) ? image[k < 64 * kRows ? k : 0] : 0.0;
        segsum += seg[i];
      }
      const double incl = wave_scan_incl(segsum, lane_v);
      double run = incl - segsum;
      const double pivot = total * (85.0f / 100.0);
      int below = 0;
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        const int k = kFirstBin + 12 * lane_v + i;
        run += seg[i];
        below += (k <= kLastBin && run < pivot) ? 1 : 0;
      }
      below = wave_sum_i(below);
      int cnt = (pivot > 0.0) ? below + 1 : 0;
      if (cnt > kBinCount) cnt = kBinCount;
      if (lane_v == 0) tmp[7] = (double)cnt;
    }
    // the next frame's image, asked for once this frame's has been read for the last time (the chunk's last frame asks for
    // its own again: no branch around the DMA)
    auto next_image = [&]() {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      dma_frame(cur + ((fi + 1 < ch.nframes) ? kHalf : 0), lane_v, nat_lds);
    };
    if (!(flags & (kBandsFeatures | kBandsFlux))) { next_image(); continue; }

    // ---- spectral_flux: Pearson r with the previous frame over bins 1..738 (SA:1919-1933,
    //      Statistics.cpp:604-638) ----
    double flux_a = 0.0, flux_aa = 0.0, flux_ab = 0.0;   // this lane's share of the whole-range sums
    if (flags & kBandsFlux) {
      double fa = 0.0, faa = 0.0, fab = 0.0;
#pragma unroll
      for (int r = 0; r < kRows; ++r) {
        const bool edge = r == 0 || r == kRows - 1;
        const mask64 ok = (r == 0) ? kFirstRowOk : kLastRowOk;
        fa += edge ? keep_where(x[r], ok) : x[r];
        faa += edge ? keep_where(xx[r], ok) : xx[r];
        fab += edge ? keep_where(xy[r], ok) : xy[r];
      }
      // (x / 738 as x * (1 / 738), the quotient through fast_div and the root through mag_sqrt: each within an ulp of
      // the generic expansions, which are three to five times the instructions)
      if (!(flags & kBandsFeatures)) {
        fa = wave_sum(fa); faa = wave_sum(faa); fab = wave_sum(fab);
        const double n = (double)kBinCount, inv_n = 1.0 / (double)kBinCount;
        const double ma = fa * inv_n, mb = fb * inv_n;
        const double denom2 = (faa - ma * ma * n) * (fbb - mb * mb * n);
        const double num = fab - (ma * mb * n);
        if (lane_v == 0) a.rec[f * a.lay.stride + a.lay.flux] = (fabs(denom2) > (double)1e-12f) ? fast_div(num, pearson_root(denom2)) : 0.0;
        fb = fa; fbb = faa;
#pragma unroll
        for (int r = 0; r < kRows; ++r) y[r] = x[r];
        next_image();
        continue;
      }
      flux_a = fa; flux_aa = faa; flux_ab = fab;
    }

    // ---- masked per-band sums: lane L ends up with band (L >> 2) & 15 ----
    const double sx = band_sum([&](int r) { return x[r]; }, lane_v, flux_a);
    const double sxx = band_sum([&](int r) { return xx[r]; }, lane_v, flux_aa);
    const double sxy = band_sum([&](int r) { return xy[r]; }, lane_v, flux_ab);
    // geometric mean: sum of log(|x| + 1e-20) (Statistics.cpp:417-455 keeps a running product and
    // takes logs only when it leaves [1e-64, 1e64]; same value up to rounding)
    // the neighbours of this lane's bins in the unsorted spectrum, for the local maxima below (they may reach into the
    // adjacent band; bin 0 and 1023 never count): asked for here, all at once, and used behind the logarithms -- loaded
    // where they are compared, each was a cache round trip of its own in a branch
    double left[kRows], right[kRows];
    // Eight logarithms per lane, not twelve: rows 5, 6 lie wholly inside band 12 and rows 8, 9, 10 (and 48 lanes of row 11)
    // inside band 13, so a lane's factors of those rows belong to one band and are multiplied first -- what the reference
    // does anyway (a running product, logarithms only when it leaves [1e-64, 1e64]).  Factors are >= 1e-20: a product of
    // four stays above 1e-80.
    double lg[8];
    {
      double fct[kRows];
#pragma unroll
      for (int r = 0; r < kRows; ++r) fct[r] = fabs(x[r]) + 1e-20;
#pragma unroll
      for (int r = 0; r < 5; ++r) lg[r] = band_log(fct[r], logt);                       // rows 0..4: several bands per row
      lg[5] = band_log(fct[5] * fct[6], logt);                                           // band 12
      lg[6] = band_log(fct[7], logt);                                                    // row 7: bands 12 | 13
      lg[7] = band_log((fct[8] * fct[9]) * (fct[10] * keep_or_one(fct[11], sub_mask(13, 11))), logt);   // band 13
    }
    double slog;
    {
      // slot -> (band, row) pairs as in band_sum; the merged slots are whole-band values
      double acc[16];
#pragma unroll
      for (int b = 0; b < 16; ++b) acc[b] = 0.0;
#pragma unroll
      for (int b = 0; b < 6; ++b) acc[b] = keep_where(lg[0], sub_mask(b, 0));
      acc[6] = keep_where(lg[0], sub_mask(6, 0)) + keep_where(lg[1], sub_mask(6, 1));
      acc[7] = keep_where(lg[1], sub_mask(7, 1));
      acc[8] = keep_where(lg[1], sub_mask(8, 1));
      acc[9] = keep_where(lg[1], sub_mask(9, 1)) + keep_where(lg[2], sub_mask(9, 2));
      acc[10] = keep_where(lg[2], sub_mask(10, 2)) + keep_where(lg[3], sub_mask(10, 3));
      acc[11] = keep_where(lg[3], sub_mask(11, 3)) + keep_where(lg[4], sub_mask(11, 4));
      acc[12] = (keep_where(lg[4], sub_mask(12, 4)) + lg[5]) + keep_where(lg[6], sub_mask(12, 7));
      acc[13] = keep_where(lg[6], sub_mask(13, 7)) + lg[7];
      static_assert(kSubStart[12] <= 320 && kSubStart[13] >= 448 && kSubStart[13] <= 512 && kSubStart[14] >= 704 && kSubStart[14] <= 768 &&
                    kSubStart[6] <= 63 && kSubStart[7] > 64 && kSubStart[9] <= 127 && kSubStart[10] > 128 && kSubStart[10] <= 191 &&
                    kSubStart[11] > 192 && kSubStart[11] <= 255 && kSubStart[12] > 256, "the log slots follow the band layout");
      slog = wave_sum16(acc, lane_v);
    }
    const double bmax = band_max([&](int r) { return x[r]; }, lane_v);

    // ---- complexity: strict local maxima above 0.25 * band maximum (SA:2170-2197); counted on the scalar
    //      unit: ballot of the peak flags of a row, masked per band, popcount ----
    wave_lds_fence();
    if ((lane_v & 3) == 0) thr[lane_v >> 2] = bmax * 0.25;
    // the neighbour bins from the image, where they are compared (bin 768, the right neighbour of a bin outside every
    // band, is not in the image: any value serves)
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
      const int k = 64 * r + lane_v;
      left[r] = image[(r == 0) ? k - (k > 0 ? 1 : 0) : k - 1];
      right[r] = image[(r == kRows - 1) ? (k + 1 < 64 * kRows ? k + 1 : k) : k + 1];
    }
    wave_lds_fence();
    int peaks[kNumSub];
#pragma unroll
    for (int b = 0; b < kNumSub; ++b) peaks[b] = 0;
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
      const int k = 64 * r + lane_v;
      const double t = thr[bid[r]];
      // bins outside every band: bin 0 (row 0) and 752..767 (last row)
      const bool inside = (r == 0) ? (lane_v >= kSubStart[0]) : ((r == kRows - 1) ? (k < kSubStart[kNumSub]) : true);
      const mask64 pk = __ballot(inside & (t > 0.0) & (x[r] > t) & (x[r] > left[r]) & (x[r] > right[r]));
#pragma unroll
      for (int b = 0; b < kNumSub; ++b)
        if (sub_touches(b, r)) peaks[b] += __popcll(pk & sub_mask(b, r));
    }
    // hand every lane_v the count of its band ((lane_v >> 2) & 15) now: the scalar counters die before the sort
    // (v_writelane: the lane that parks band b's sums, 4 b, gets its count -- a select chain over the bands was 43 instructions)
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < kNumSub; ++i) asm("v_writelane_b32 %0, %1, %2" : "+v"(cnt) : "s"(peaks[i]), "n"(4 * i));
    // the image has been read for the last time (the ballots above consumed the neighbours): the next frame's travels
    // while the keys are sorted and the cuts summed
    next_image();

    // ---- contrast: sort (band, value) keys to find the key at each band's two cuts, then exact sums ----
    // The keys go into bin order (position p = bin: lane p / 16, register p % 16) and each block of 256 positions is
    // sorted on its own (the levels K <= 256 of the network pair positions of one block only): band is the major part of
    // a key and the bands are runs of consecutive bins, so bands 0..10 and 12 come out sorted inside blocks 0 and 1; only
    // band 11 (bins 221..316) and band 13 (465..751) lie across a block boundary, as two sorted runs each, and their four
    // cut keys are order statistics of two sorted runs: a bisection.  The levels K = 512, 1024 of the full sort -- 11
    // cross-lane and 8 in-lane stages, 830 of its 1 900 instructions -- are not run.
    u32 key[16];
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < kRows; ++r) sorted[64 * r + lane_v] = sort_key(bid[r], x[r]);
    wave_lds_fence();
    {
      const uint4* const mine = reinterpret_cast<const uint4*>(sorted + 16 * lane_v);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint4 q4 = mine[i];
        key[4 * i] = q4.x; key[4 * i + 1] = q4.y; key[4 * i + 2] = q4.z; key[4 * i + 3] = q4.w;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) key[i] = (lane_v < 4 * kRows) ? key[i] : 0x7FFFFFFFu;   // positions 768..1023: nothing
    }
    sort_level<256>(key, lane_v);
    wave_lds_fence();
#pragma unroll
    for (int i = 0; i < 16; ++i) sorted[16 * lane_v + i] = key[i];
    wave_lds_fence();
    // Where the bands sit now.  Block 0 = bins 0..255: bands 0..10 at [start - 1, ..) (bin 0, outside every band, has the
    // largest key of the block: position 255), the first 35 bins of band 11 at [220, 255).  Block 1 = bins 256..511: the
    // other 61 bins of band 11 at [256, 317), band 12 at [317, 465), the first 47 bins of band 13 at [465, 512).  Block 2:
    // the other 240 bins of band 13 at [512, 752).
    // The four cut keys (ranks nn - 1, nn, n - nn - 1, n - nn) of band 11 (sel[0..3]) and band 13 (sel[4..7]): the key of
    // rank t of two sorted runs A = [a0, a0 + na), B = [b0, b0 + nb) is max(A[i - 1], B[t - i]) where i, the number of
    // elements of A among the t + 1 smallest, is the smallest i with B[t - i] <= A[i].  The predicate B[t - i] > A[i] is
    // true below that i and false from it on, so i = lo + (number of candidates in [lo, hi) for which it holds): every
    // candidate is tested by a lane of its own, one ballot per rank -- a bisection by eight lanes was six dependent LDS
    // round trips.
    {
      int at[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const bool b13 = c >= 4;
        const int a0 = b13 ? 465 : 220, na = b13 ? 47 : 35, b0 = b13 ? 512 : 256, nb = b13 ? 240 : 61;
        const int n = na + nb, nn = b13 ? kSubNeigh[13] : kSubNeigh[11], which = c & 3;
        const int t = (which == 0) ? nn - 1 : (which == 1 ? nn : (which == 2 ? n - nn - 1 : n - nn));
        const int lo = (t + 1 - nb > 0) ? t + 1 - nb : 0, hi = (t + 1 < na) ? t + 1 : na;   // hi - lo <= 48 candidates
        // (the positions read stay inside the array for every lane: no guard on the reads)
        const bool more = sorted[b0 + t - lo - lane_v] > sorted[a0 + lo + lane_v];
        at[c] = lo + __popcll(__ballot(more) & ((1ull << (hi - lo)) - 1ull));
      }
      u32 found[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const bool b13 = c >= 4;
        const int a0 = b13 ? 465 : 220, na = b13 ? 47 : 35, b0 = b13 ? 512 : 256, nb = b13 ? 240 : 61;
        const int n = na + nb, nn = b13 ? kSubNeigh[13] : kSubNeigh[11], which = c & 3;
        const int t = (which == 0) ? nn - 1 : (which == 1 ? nn : (which == 2 ? n - nn - 1 : n - nn));
        const int i = at[c];                           // wave-uniform, lo <= i <= hi: i - 1 < na, t - i < nb
        const u32 ka = sorted[a0 + (i >= 1 ? i - 1 : 0)], kb = sorted[b0 + (t - i >= 0 ? t - i : 0)];
        const u32 ka0 = (i >= 1) ? ka : 0u, kb0 = (t - i >= 0) ? kb : 0u;
        found[c] = ka0 > kb0 ? ka0 : kb0;
      }
      if (lane_v == 0) {
#pragma unroll
        for (int c = 0; c < 8; ++c) sel[c] = found[c];
      }
    }
    wave_lds_fence();
    mask64 ties_v, ties_p;   // bands whose cut goes through a class of equal keys
    {
      const int bl = lane_v & 15;
      const int pv = cutpos[bl], pp = cutpos[16 + bl];   // positions of the two cut keys in the sorted blocks (LDS table)
      const bool band_lane = lane_v < kNumSub;
      u32 kv = sorted[pv], kv_next = sorted[pv + 1], kp = sorted[pp], kp_prev = sorted[pp - 1];
      const bool two_runs = bl == 11 || bl == 13;
      const int s0 = (bl == 13) ? 4 : 0;
      kv = two_runs ? sel[s0] : kv;
      kv_next = two_runs ? sel[s0 + 1] : kv_next;
      kp_prev = two_runs ? sel[s0 + 2] : kp_prev;
      kp = two_runs ? sel[s0 + 3] : kp;
      // a bin outside every band (bid 15) meets cut keys 0 / ~0: never below the one, never above the other
      if (lane_v < 16) {
        cut[lane_v] = band_lane ? kv : 0u;
        cut[16 + lane_v] = band_lane ? kp : 0xFFFFFFFFu;
      }
      // the valley's cut goes through a tie class when the bin right after it has the same key (nn < n always);
      // the peak's when the bin right before it has
      ties_v = __ballot(band_lane && kv_next == kv);
      ties_p = __ballot(band_lane && kp_prev == kp);
    }
    wave_lds_fence();
    double vsum, psum;
    {
      double xs[kRows];
#pragma unroll
      for (int r = 0; r < kRows; ++r) xs[r] = (sort_key(bid[r], x[r]) <= cut[bid[r]]) ? x[r] : 0.0;
      vsum = band_sum([&](int r) { return xs[r]; }, lane_v);
#pragma unroll
      for (int r = 0; r < kRows; ++r) xs[r] = (sort_key(bid[r], x[r]) >= cut[16 + bid[r]]) ? x[r] : 0.0;
      psum = band_sum([&](int r) { return xs[r]; }, lane_v);
    }
    // cuts inside a tie class: exact resolution, band by band (wave-uniform loops; rare)
    for (mask64 m = ties_v; m; m &= m - 1) {
      const int bb = __ffsll((unsigned long long)m) - 1;
      const double sv = exact_cut_sum<true>(cur, bb, cut[bb], lane_v);
      vsum = (((lane_v >> 2) & 15) == bb) ? sv : vsum;
    }
    for (mask64 m = ties_p; m; m &= m - 1) {
      const int bb = __ffsll((unsigned long long)m) - 1;
      const double sp = exact_cut_sum<false>(cur, bb, cut[16 + bb], lane_v);
      psum = (((lane_v >> 2) & 15) == bb) ? sp : psum;
    }

    // ---- park the frame's sums in its slot of the group (LDS): the per-band sums of up to four frames wait for their
    //      closed forms, and the quotients, roots, logarithms and exponentials of a group are evaluated once, in all 64
    //      lanes (lane L: band L >> 2 of frame L & 3 of the group), when it is full or the chunk ends -- instead of once per
    //      frame in the 14 lanes that matter ----
    {
      const int slot = fi & 3;
      if ((lane_v & 3) == 0) {
        double* const mine = park + slot * (kParked * 16) + (lane_v >> 2);
        mine[0 * 16] = sx;   mine[1 * 16] = sxx;  mine[2 * 16] = sxy;  mine[3 * 16] = slog; mine[4 * 16] = sy;
        mine[5 * 16] = syy;  mine[6 * 16] = vsum; mine[7 * 16] = psum; mine[8 * 16] = (double)cnt;
      }
      if (slot == 3 || fi == ch.nframes - 1) {
        wave_lds_fence();
        const double* const mine = park + (lane_v & 3) * (kParked * 16) + (lane_v >> 2);
        const double p_sx = mine[0 * 16], p_sxx = mine[1 * 16], p_sxy = mine[2 * 16], p_slog = mine[3 * 16], p_sy = mine[4 * 16];
        const double p_syy = mine[5 * 16], p_vsum = mine[6 * 16], p_psum = mine[7 * 16], p_cnt = mine[8 * 16];
        // ---- per-band results: lane L finishes band L >> 2 of frame (f - slot) + (L & 3) ----
        const int b = lane_v >> 2, j = lane_v & 3;
        const bool live = j <= slot;
        // 1 / n and 1 / neighbours of this lane's band: from the wave's LDS table (a select chain per frame was 60 instructions)
        const double nb = cn[2 * 16 + b], inv_nb = cn[b], inv_nn = cn[16 + b];
        // quotients by the band's bin count as products with its reciprocal, the others through fast_div (exact when the
        // quotient is representable, one ulp otherwise), roots through mag_sqrt: the generic expansions of this stage were
        // 17 divisions and 2 roots of ~25 instructions each
        const double mean = p_sx * inv_nb;                                 // TStatistics::Mean (n >= 2 for every band)
        const double rms = mag_sqrt(p_sxx * inv_nb);                       // SA:2154-2159
        const double gm = fast_exp(p_slog * inv_nb);                       // Statistics.cpp:442
        const double fl = (mean == 0.0) ? 0.0 : fast_div(gm, mean);        // Statistics.cpp:565-574
        double fdb = lin_to_db(fl) * (-1.0 / 60.0);                        // SFlatnessDb, SA:129-133
        fdb = fdb < 1.0 ? fdb : 1.0;
        const double ma = mean, mb = p_sy * inv_nb;                        // Statistics.cpp:604-638
        const double denom2 = (p_sxx - ma * ma * nb) * (p_syy - mb * mb * nb);
        const double num = p_sxy - (ma * mb * nb);
        const double flux = (fabs(denom2) > (double)1e-12f) ? fast_div(num, pearson_root(denom2)) : 0.0;
        const double valley = p_vsum * inv_nn + 1e-30, peakv = p_psum * inv_nn + 1e-30;  // SA:2216, 2228
        // pow(a, b) = exp(b log a), a > 0 (SA:2231-2232)
        // (the exponent's quotient stays the generic one: a band mean of exactly 1.0 makes its denominator 0)
        const double contrast = -1.0 * fast_exp(fast_log(fast_div(peakv, valley)) / fast_log(mean + 1e-30));

        double* const rec = a.rec + (f - slot + j) * a.lay.stride;
        if (live && b < kNumSub) {
          rec[a.lay.sub_rms + b] = rms;
          rec[a.lay.sub_flat + b] = fdb;
          rec[a.lay.sub_flux + b] = flux;
          rec[a.lay.sub_cplx + b] = p_cnt;
          rec[a.lay.sub_contrast + b] = contrast;
        }
        // spectral_flux (SA:1919-1933, Statistics.cpp:604-638): the same expression over the whole analysis range
        if (live && b == kWholeBand && (flags & kBandsFlux)) rec[a.lay.flux] = flux;
        // spectral_contrast = mean of the 14 band contrasts, summed in band order (SA:2252-2260)
        double csum = 0.0;
#pragma unroll
        for (int i = 0; i < kNumSub; ++i) csum += __shfl(contrast, 4 * i + j);
        if (live && b == 0) rec[a.lay.contrast] = csum / (double)kNumSub;
      }
    }

    // this frame is the next one's predecessor
    sy = sx; syy = sxx;
#pragma unroll
    for (int r = 0; r < kRows; ++r) y[r] = x[r];
  }
  }
}

}  // namespace

hipError_t launch_bands(const BandArgs& a, hipStream_t stream) {
  if (a.n_chunks <= 0) return hipSuccess;
  const int want = (a.n_chunks + 3) / 4;
  constexpr int kAll = kBandsFeatures | kBandsFlux;
  auto launch = [&](auto kernel) {
    // with a work queue: the workgroups that are resident at once, each wave drawing chunks until there are none
    static const int resident = resident_blocks(kernel, 256, 0);
    const int cap = a.queue.counter ? resident : 256 * 16;
    hipLaunchKernelGGL(kernel, dim3(want < cap ? want : cap), dim3(256), 0, stream, a);
  };
  if (a.flags == kAll) launch(bands_kernel<kAll>);
  else if (a.flags == (kAll | kBandsSpectrum)) launch(bands_kernel<kAll | kBandsSpectrum>);
  else if (a.flags == (kAll | kBandsSpectrum | kBandsStats)) launch(bands_kernel<kAll | kBandsSpectrum | kBandsStats>);
  else launch(bands_kernel<-1>);
  return hipGetLastError();
}

}  // namespace afx
