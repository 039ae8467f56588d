// afec_amd/csrc/afx_kernels.hip -- gfx950 kernels of the low-level spectral hot path.
//
// One wavefront (64 lanes) owns a run of consecutive frames of one buffer ("chunk").  Per frame it
// computes the 2048-point FFT of the windowed real frame as a 1024-point complex FFT of
// z[n] = x[2n] + i x[2n+1] held as 16 complex values per lane:
//
//   load      v[r]  = z[64 r + lane]                 (hop reuse: rows 0..7 are last frame's 8..15)
//   P1        16-point DFT over r, in registers      -> j1        lane = 16 m2 + n2
//   E1        register transpose with v_permlane32_swap / v_permlane16_swap: lane bits 5,4 (m2)
//             trade places with register bits 3,2 of j1 -> lane = 16 jh + n2, reg = 4 m2 + jl
//             (j1 = 4 jh + jl).  No LDS traffic.
//   T1        * w64^(m2 j1)
//   P2        4-point DFT over m2, in registers      -> j2        reg = 4 j2 + jl
//   T2        * w1024^(n2 (j1 + 16 j2))
//   E2        LDS exchange  -> (lane''= k1 = 16 j2 + j1, reg = n2), slot = k1 + 65 n2
//   P3        16-point DFT over n2                   -> Z[lane + 64 k2]
//   E3        partner Z[1024-k] by cross-lane read, even/odd untangle, |X[k]| for k = lane + 64 r
//
// then the descriptors straight from the magnitudes in registers.  The index algebra and the LDS
// map (conflict-free for ds_write_b64 / ds_read_b64, address = lane part + immediate) are
// modelled and checked in tools/fft_dataflow_model.py.
//
// A workgroup is kWaves independent waves that share the constant tables (twiddles, mel rows, DCT
// basis) in LDS; each wave has a private 8.1 KiB exchange plane.  Waves never synchronise with
// each other after the table load.  The kernel is bound by the CU's LDS instruction issue
// (profiles/r01), which is why E1 lives in the VALU and the double kernels read the window from
// global memory (L1/L2-resident) rather than from LDS.
//
// What each stage replaces in the reference (SampleAnalyser.cpp = SA):
//   window+FFT+magnitude  SA:826-845 (xtract_windowed, TFftTransformComplex, TAudioMath::Magnitude)
//   mel+log+DCT           SA:2052-2063 -> LibXtract vector.c:350-391
//   rms/centroid/spread/skew/kurt/rolloff/flatness        SA:1808-1915 -> Statistics.cpp, scalar.c
//   (flux, SA:1919-1933, needs the previous frame: afx_bands.hip, from the stored magnitudes)
//   28 bands              SA:2007-2048
//   amplitude peak/rms    SA:1760-1783

#include <hip/hip_runtime.h>

#include "afx_internal.h"
#include "afx_device.h"
#include "afx_fft.h"

namespace afx {
namespace {


// feature classes the kernel is specialised for (the host picks the smallest that covers the mask)
constexpr int kFeatC2 = 0;     // MFCC only: magnitudes of bins 0..383
constexpr int kFeatStats = 1;  // + rms/centroid/spread/skew/kurt/rolloff/flatness, amplitude: bins 0..767
constexpr int kFeatFull = 2;   // + 28 bands, magnitude output: all 1024 bins (flux and the sub-band
                               // descriptors are computed from the stored magnitudes by afx_bands.hip)

template <typename TIn>
struct InPair;
template <>
struct InPair<float> {
  using type = float2;
};
template <>
struct InPair<double> {
  using type = double2;
};

// ---- LDS layout: shared tables, then one exchange plane per wave ----
// LDS layout: shared tables, then one exchange plane per wave.  The double kernels read the
// window from global memory (16 KiB, L1/L2-resident, coalesced 16 B per lane): the kernel is bound by
// LDS instruction issue, the vector-memory path is otherwise idle.
template <typename T, int POST_ROWS>
struct LdsMap {
  static constexpr bool win_global = sizeof(T) == 8;
  static constexpr int win = 0;                                                     // [16][64] cx<T>
  static constexpr int t2 = win + (win_global ? 0 : 1024 * (int)sizeof(cx<T>));     // [16][64] cx<T>
  static constexpr int post = t2 + 1024 * (int)sizeof(cx<T>);                       // [POST_ROWS][64] cx<T>
  static constexpr int melw = post + POST_ROWS * 64 * (int)sizeof(cx<T>);           // [22][64] T
  static constexpr int dct = melw + kMelPairs * 64 * (int)sizeof(T);                // [14][16] double
  static constexpr int xchg = dct + 14 * 16 * 8;                                    // kWaves planes
  static constexpr int plane_bytes = kPlaneSlots * 8;
  static constexpr int total(int waves) { return xchg + waves * plane_bytes; }
};

__device__ __forceinline__ void copy_to_lds(unsigned char* dst, const void* src, int bytes, int tid, int nthreads) {
  const uint4* s = reinterpret_cast<const uint4*>(src);
  uint4* d = reinterpret_cast<uint4*>(dst);
  for (int i = tid; i < bytes / 16; i += nthreads) d[i] = s[i];
}

// log + 14-point DCT-II + store for up to four frames whose mel sums sit in lanes 4 f + slot
__device__ __forceinline__ void finish_mfcc(double acc, int nslots, int64_t row0, const FrameArgs& a,
                                            const double* dct, int lane) {
  const double lg = log(acc < 2e-42 ? 2e-42 : acc);  // XTRACT_LOG_LIMIT, vector.c:364
  const int n = (lane >> 2) & 15, s = lane & 3;
  const double* drow = dct + 16 * (n < kNumCep ? n : 0);
  double c = 0.0;
#pragma unroll
  for (int m = 0; m < kNumCep; ++m) c += __shfl(lg, 4 * m + s) * drow[m];
  if (n < kNumCep && s < nslots) a.rec[(row0 + s) * a.lay.stride + a.lay.mfcc + n] = c;
}

template <typename T, typename TIn, int FEAT, int WAVES, bool SCALED>
__global__ __launch_bounds__(WAVES * 64) void frames_kernel(const FrameArgs a) {
  using Pair = typename InPair<TIn>::type;
  constexpr int MR = (FEAT == kFeatC2) ? kMelRows : (FEAT == kFeatStats ? 12 : 16);
  using Map = LdsMap<T, MR>;
  extern __shared__ __align__(16) unsigned char lds_raw[];
  const int lane0 = threadIdx.x & 63;
  const int lane = lane0;
  const int wave = threadIdx.x >> 6;

  // ---- shared tables ----
  if (!Map::win_global) copy_to_lds(lds_raw + Map::win, a.win, 1024 * (int)sizeof(cx<T>), threadIdx.x, WAVES * 64);
  copy_to_lds(lds_raw + Map::t2, a.t2, 1024 * (int)sizeof(cx<T>), threadIdx.x, WAVES * 64);
  copy_to_lds(lds_raw + Map::post, a.post, MR * 64 * (int)sizeof(cx<T>), threadIdx.x, WAVES * 64);
  copy_to_lds(lds_raw + Map::melw, a.melw, kMelPairs * 64 * (int)sizeof(T), threadIdx.x, WAVES * 64);
  copy_to_lds(lds_raw + Map::dct, a.dct, 14 * 16 * 8, threadIdx.x, WAVES * 64);
  __syncthreads();

  const cx<T>* const win = (Map::win_global ? reinterpret_cast<const cx<T>*>(a.win)
                                            : reinterpret_cast<const cx<T>*>(lds_raw + Map::win)) + lane;
  const cx<T>* const t2 = reinterpret_cast<const cx<T>*>(lds_raw + Map::t2) + lane;
  const cx<T>* const post = reinterpret_cast<const cx<T>*>(lds_raw + Map::post) + lane;
  // T1 table [jh][m2][jl] = w64^(m2 (4 jh + jl)): 12 values per lane, the same for the 16 lanes of
  // a row -> one cache line per row from L1
  const cx<T>* const t1 = reinterpret_cast<const cx<T>*>(a.t1) + 16 * (lane >> 4);
  const T* const melw = reinterpret_cast<const T*>(lds_raw + Map::melw) + lane;
  const double* const dct = reinterpret_cast<const double*>(lds_raw + Map::dct);
  unsigned char* const plane = lds_raw + Map::xchg + wave * Map::plane_bytes;
  double* const lds_mag = reinterpret_cast<double*>(plane);
  // 32-bit LDS byte address of the plane for the asm reads
  const unsigned plane_addr =
      (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds_raw + Map::xchg + wave * Map::plane_bytes;

  // E2 lane parts: this lane holds (jh, n2) when writing and is k1 when reading
  const int e2w = 4 * (lane >> 4) + 65 * (lane & 15);   // + 16 j2 + jl
  const unsigned e2r_addr = plane_addr + 8u * lane;     // + 8 * 65 n2
  const int partner = (64 - lane) & 63;

  const TIn* const pcm = reinterpret_cast<const TIn*>(a.pcm);
  const int wave_global = blockIdx.x * WAVES + wave;
  const int wave_stride = gridDim.x * WAVES;

  for (int ci = wave_global; ci < a.n_chunks; ci += wave_stride) {
    const Chunk ch = a.chunks[ci];
    const Pair* src = reinterpret_cast<const Pair*>(pcm + ch.sample_off) + lane;
    const int total = ch.nframes;
    const double sc = SCALED ? wave_uniform(ch.scale) : 1.0;   // SCALED: the buffer's FinalScaling (the arena holds LoadSample's float signal)

    Pair lo[8], nxt[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) lo[r] = src[64 * r];
#pragma unroll
    for (int r = 0; r < 8; ++r) nxt[r] = src[64 * (r + 8)];

    double mel_acc = 0.0;  // mel sums of up to four finished frames: lane 4 f + slot
    int pending = 0;
    int64_t pending_row0 = 0;

    for (int fi = 0; fi < total; ++fi) {
      // the statistics classes test hundreds of lane ranges: re-materialise the lane id per frame so the
      // predicates are recomputed instead of being hoisted into (spilled) SGPR pairs
      int lane = lane0;
      if (FEAT != kFeatC2) asm volatile("" : "+v"(lane));
      // ---- time-domain descriptors on the hop = rows 0..7 (SA:871-872) ----
      double amp_peak = 0.0, amp_sq = 0.0;
      if (FEAT != kFeatC2 && (a.mask & ((1u << 11) | (1u << 12)))) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const double x0 = pcm_double<SCALED>(lo[r].x, sc), x1 = pcm_double<SCALED>(lo[r].y, sc);
          amp_peak = fmax(amp_peak, fmax(fabs(x0), fabs(x1)));
          amp_sq += x0 * x0 + x1 * x1;
        }
        amp_peak = wave_max(amp_peak);
        amp_sq = wave_sum(amp_sq);
      }

      // ---- window (table carries the 1/2048 of kDivFwdByN and the 1/2 of the untangle) ----
      cx<T> v[16];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const cx<T> w0 = win[64 * r];
        const cx<T> w1 = win[64 * (r + 8)];
        v[r] = {(T)pcm_double<SCALED>(lo[r].x, sc) * w0.re, (T)pcm_double<SCALED>(lo[r].y, sc) * w0.im};
        v[r + 8] = {(T)pcm_double<SCALED>(nxt[r].x, sc) * w1.re, (T)pcm_double<SCALED>(nxt[r].y, sc) * w1.im};
        lo[r] = nxt[r];
      }
      // prefetch the next frame's new hop (rows 8..15 of frame fi + 1)
      if (fi + 1 < total) {
        const Pair* nsrc = src + (size_t)(fi + 1) * (kHop / 2);
#pragma unroll
        for (int r = 0; r < 8; ++r) nxt[r] = nsrc[64 * (r + 8)];
      }

      // ---- P1, E1 (register transpose), T1 ----
      dft16(v);
      transpose_m2_into_registers(v);   // lane = 16 jh + n2, v[4 m2 + jl]
#pragma unroll
      for (int g = 4; g < 16; ++g) v[g] = cmul(v[g], t1[g]);

      // ---- P2 + T2 + E2 ----
#pragma unroll
      for (int jl = 0; jl < 4; ++jl) radix4(v[jl], v[4 + jl], v[8 + jl], v[12 + jl]);
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        v[g] = cmul(v[g], t2[64 * g]);
      }
      Xchg<T>::run(plane, e2r_addr, e2w, v);

      // ---- P3: v[k2] = Z[lane + 64 k2] ----
      dft16(v);

      // ---- E3 + untangle + magnitude: mag[r] = |X[lane + 64 r]| ----
      T mag[MR];
#pragma unroll
      for (int r = 0; r < MR; ++r) {
        cx<T> p{__shfl(v[15 - r].re, partner), __shfl(v[15 - r].im, partner)};
        if (lane == 0) p = v[(16 - r) & 15];
        const cx<T> z = v[r];
        const cx<T> w = post[64 * r];
        const T er = z.re + p.re, ei = z.im - p.im;   // E = Z + conj(P)
        const T orr = z.im + p.im, oi = p.re - z.re;  // O = -i (Z - conj(P))
        const T xr = er + (w.re * orr - w.im * oi);
        const T xi = ei + (w.re * oi + w.im * orr);
        mag[r] = mag_sqrt(xr * xr + xi * xi);
      }

      const bool emit = true;
      const int64_t row = (int64_t)ch.frame0 + fi;
      double* const rec = a.rec + row * a.lay.stride;

      if (emit) {
        if (FEAT == kFeatFull && a.mag_out) {
#pragma unroll
          for (int r = 0; r < MR; ++r) a.mag_out[row * kHalf + 64 * r + lane] = (double)mag[r];
        }

        // ---- MFCC: sparse mel rows now; log + DCT once per four frames (vector.c:350-391) ----
        if (FEAT == kFeatC2 || (a.mask & 1u)) {
          // mel partial sums in the kernel's own precision (the float kernels' magnitudes carry 1e-7
          // already); the log and the DCT below are always double
          T e[16];
#pragma unroll
          for (int f = 0; f < 16; ++f) e[f] = (T)0;
#pragma unroll
          for (int r = 0; r < kMelRows; ++r)
#pragma unroll
            for (int f = 0; f < kNumCep; ++f)
              if (mel_touches(f, r)) e[f] += mag[r] * melw[64 * mel_pair_index(r, f)];
          const double tot = (double)wave_sum16(e, lane);  // lane L: filter (L >> 2) & 15
          if (pending == 0) pending_row0 = row;
          if ((lane & 3) == pending) mel_acc = tot;
          if (++pending == 4) {
            finish_mfcc(mel_acc, 4, pending_row0, a, dct, lane);
            pending = 0;
          }
        }

        // ---- spectral statistics over bins 1..738, j = bin - 1 (SA:1808-1933) ----
        if (FEAT != kFeatC2 && (a.mask & 0xFEu)) {
          double s1 = 0.0, s2 = 0.0, sj = 0.0, prod = 1.0;
#pragma unroll
          for (int r = 0; r < 12; ++r) {
            const int k = 64 * r + lane;
            const bool ok = (r == 0) ? (lane >= kFirstBin) : (r == 11 ? (k <= kLastBin) : true);
            const double m = ok ? (double)mag[r] : 0.0;
            s1 += m;
            s2 += m * m;
            sj += (double)(k - kFirstBin) * m;
            prod *= ok ? (m + 1e-20) : 1.0;
          }
          s1 = wave_sum(s1);
          // x / 738 as x * (1 / 738): within an ulp of the reference's division, a tenth of its instructions
          const double inv_n = 1.0 / (double)kBinCount;
          if (a.mask & (1u << 1)) {
            s2 = wave_sum(s2);
            if (lane == 0) rec[a.lay.srms] = nan_to_zero(sqrt(s2 * inv_n));
          }
          double cen = 0.0, spr = 0.0;
          if (a.mask & 0x3Cu) {  // centroid, spread, skewness, kurtosis
            sj = wave_sum(sj);
            if (s1 != 0.0) {
              cen = fast_div(sj, s1);
              double sv = 0.0;
#pragma unroll
              for (int r = 0; r < 12; ++r) {
                const int k = 64 * r + lane;
                const bool ok = (r == 0) ? (lane >= kFirstBin) : (r == 11 ? (k <= kLastBin) : true);
                const double t = (double)(k - kFirstBin) - cen;
                sv += ok ? t * t * (double)mag[r] : 0.0;
              }
              spr = fast_div(wave_sum(sv), s1);
            }
            if (lane == 0) {
              if (a.lay.centroid >= 0) rec[a.lay.centroid] = cen;
              if (a.lay.spread >= 0) rec[a.lay.spread] = spr;
            }
            if (a.mask & 0x30u) {
              double sk = 0.0, ku = 0.0;
              if (fabs(spr) > (double)1e-12f) {
                const double inv = fast_div(1.0, spr);
#pragma unroll
                for (int r = 0; r < 12; ++r) {
                  const int k = 64 * r + lane;
                  const bool ok = (r == 0) ? (lane >= kFirstBin) : (r == 11 ? (k <= kLastBin) : true);
                  const double t = ((double)mag[r] - cen) * inv;
                  const double tt = t * t;
                  sk += ok ? tt * t : 0.0;
                  ku += ok ? tt * tt : 0.0;
                }
                sk = wave_sum(sk) * inv_n;
                ku = wave_sum(ku) * inv_n - 3.0;
              }
              if (lane == 0) {
                if (a.lay.skew >= 0) rec[a.lay.skew] = sk;
                if (a.lay.kurt >= 0) rec[a.lay.kurt] = ku;
              }
            }
          }
          if (a.mask & (1u << 7)) {  // flatness: GM / AM in dB / -60, clamped (SA:129-133)
            const double sumlog = wave_sum(fast_log(prod));
            const double gm = fast_exp(sumlog * inv_n);
            const double am = s1 * inv_n;
            const double fl = (am == 0.0) ? 0.0 : fast_div(gm, am);
            const double d = lin_to_db(fl) * (-1.0 / 60.0);
            if (lane == 0) rec[a.lay.flatness] = nan_to_zero(d < 1.0 ? d : 1.0);
          }
          if (a.mask & (1u << 6)) {  // rolloff (scalar.c:472-492): 43 * #elements until 85 %
            // natural-order copy in LDS, then each lane walks 12 consecutive bins
            wave_lds_fence();
#pragma unroll
            for (int r = 0; r < 12; ++r) lds_mag[64 * r + lane] = (double)mag[r];
            wave_lds_fence();
            double seg[12];
            double segsum = 0.0;
#pragma unroll
            for (int i = 0; i < 12; ++i) {
              const int k = kFirstBin + 12 * lane + i;
              seg[i] = (k <= kLastBin) ? lds_mag[k] : 0.0;
              segsum += seg[i];
            }
            const double incl = wave_scan_incl(segsum, lane);
            double run = incl - segsum;
            const double pivot = s1 * (85.0f / 100.0);
            int below = 0;
#pragma unroll
            for (int i = 0; i < 12; ++i) {
              const int k = kFirstBin + 12 * lane + i;
              run += seg[i];
              below += (k <= kLastBin && run < pivot) ? 1 : 0;
            }
            below = wave_sum_i(below);
            int cnt = (pivot > 0.0) ? below + 1 : 0;
            if (cnt > kBinCount) cnt = kBinCount;  // the reference loop is unbounded; clamp at the slice end
            if (lane == 0) rec[a.lay.rolloff] = (double)cnt * (double)(kSampleRate / (kFft / 2));
          }
        }

        // ---- 28 spectrum bands: sum of squared magnitudes (SA:2007-2048) ----
        if (FEAT == kFeatFull && (a.mask & (1u << 9))) {
          double mine = 0.0;
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            double acc[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const int b = 16 * half + i;
              acc[i] = 0.0;
              if (b < kNumBands) {
#pragma unroll
                for (int r = 0; r < MR; ++r)
                  if (band_touches(b, r)) {
                    const int k = 64 * r + lane;
                    const double m = (double)mag[r];
                    acc[i] += (k >= kBandEdge[b] && k < kBandEdge[b + 1]) ? m * m : 0.0;
                  }
              }
            }
            const double tot = wave_sum16(acc, lane);  // lane L: band 16 half + ((L >> 2) & 15)
            if (((lane >> 6) == 0) && ((lane & 3) == half)) mine = tot;
          }
          // lanes with (lane & 3) == half hold band 16 half + (lane >> 2)
          const int b = 16 * (lane & 3) + (lane >> 2);
          if ((lane & 3) < 2 && b < kNumBands) rec[a.lay.bands + b] = mine;
        }

        if (FEAT != kFeatC2 && lane == 0) {
          if (a.mask & (1u << 11)) rec[a.lay.amp_peak] = amp_peak;
          if (a.mask & (1u << 12)) rec[a.lay.amp_rms] = nan_to_zero(sqrt(amp_sq / (double)kHop));
        }
      }
    }
    if (pending > 0) finish_mfcc(mel_acc, pending, pending_row0, a, dct, lane);
  }
}

// waves per workgroup (one workgroup per CU): limited by LDS (tables + 8.5 KiB per wave <= 160 KiB)
// and by the VGPR budget that many waves per SIMD leave
constexpr int kWavesC2 = 8;
constexpr int kWavesOther = 8;

template <typename T, typename TIn, int FEAT, int WAVES, bool SCALED>
hipError_t launch_one(const FrameArgs& a, int grid_blocks, hipStream_t stream) {
  constexpr int MR = (FEAT == kFeatC2) ? kMelRows : (FEAT == kFeatStats ? 12 : 16);
  const size_t lds = (size_t)LdsMap<T, MR>::total(WAVES);
  auto k = frames_kernel<T, TIn, FEAT, WAVES, SCALED>;
  static bool attribute_set[16] = {};   // per device: raising the dynamic LDS limit once is enough
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 16 || !attribute_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 16) attribute_set[dev] = true;
  }
  hipLaunchKernelGGL(k, dim3(grid_blocks), dim3(WAVES * 64), lds, stream, a);
  return hipGetLastError();
}

template <typename T, typename TIn, bool SCALED>
hipError_t launch_typed(const FrameArgs& a, int feat, int grid_blocks, hipStream_t stream) {
  switch (feat) {
    case kFeatC2: return launch_one<T, TIn, kFeatC2, kWavesC2, SCALED>(a, grid_blocks, stream);
    case kFeatStats: return launch_one<T, TIn, kFeatStats, kWavesOther, SCALED>(a, grid_blocks, stream);
    default: return launch_one<T, TIn, kFeatFull, kWavesOther, SCALED>(a, grid_blocks, stream);
  }
}

}  // namespace

int frames_feature_class(uint32_t mask) {
  if (mask == 1u) return kFeatC2;
  if (mask & ((1u << 8) | (1u << 9) | (1u << 10) | (1u << 13))) return kFeatFull;
  return kFeatStats;
}

int frames_waves_per_block(uint32_t mask) { return frames_feature_class(mask) == kFeatC2 ? kWavesC2 : kWavesOther; }

hipError_t launch_frames(const FrameArgs& a, int precision, int pcm_dtype, int grid_blocks,
                         hipStream_t stream) {
  if (a.n_chunks <= 0) return hipSuccess;
  const int feat = frames_feature_class(a.mask);
  if (precision != 0) return hipErrorInvalidValue;   // the float STFT mode is gone (include/afx.h, AFX_PRECISION_F32)
  if (pcm_dtype == kPcmScaledF32) return launch_typed<double, float, true>(a, feat, grid_blocks, stream);
  return pcm_dtype == kPcmF32 ? launch_typed<double, float, false>(a, feat, grid_blocks, stream)
                              : launch_typed<double, double, false>(a, feat, grid_blocks, stream);
}

}  // namespace afx
