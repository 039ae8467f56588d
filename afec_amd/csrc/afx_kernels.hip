// afec_amd/csrc/afx_kernels.hip -- gfx950 kernels of the low-level spectral hot path.
//
// One wavefront (64 lanes) owns a run of consecutive frames of one buffer ("chunk").  Per frame it
// computes the 2048-point FFT of the windowed real frame as a 1024-point complex FFT of
// z[n] = x[2n] + i x[2n+1] held as 16 complex values per lane:
//
//   load      v[r]  = z[64 r + lane]                 (hop reuse: rows 0..7 are last frame's 8..15)
//   P1        16-point DFT over r, in registers      -> j1        lane = 16 m2 + 4 h + q
//   T1        * w64^(m2 j1)
//   E1        LDS exchange  (lane,(j1)) -> (lane'=4 j1 + h, reg = 4 m2 + q)
//   P2        4-point DFT over m2, in registers      -> j2
//   T2        * w1024^((4h+q)(j1 + 16 j2))
//   E2        LDS exchange  -> (lane''= j1 + 16 j2, reg = 4 h + q)
//   P3        16-point DFT over n2 = 4h+q            -> Z[lane + 64 k2]
//   E3        partner Z[1024-k] by cross-lane read, even/odd untangle, |X[k]| for k = lane + 64 r
//
// then the descriptors straight from the 16 magnitudes per lane.  The index algebra and the LDS
// maps (conflict-free for ds_write_b64 / ds_read_b64, address = lane part + immediate) are
// modelled and checked in tools/fft_dataflow_model.py.
//
// What each stage replaces in the reference (SampleAnalyser.cpp = SA):
//   window+FFT+magnitude  SA:826-845 (xtract_windowed, TFftTransformComplex, TAudioMath::Magnitude)
//   mel+log+DCT           SA:2052-2063 -> LibXtract vector.c:350-391
//   rms/centroid/spread/skew/kurt/rolloff/flatness/flux   SA:1808-1933 -> Statistics.cpp, scalar.c
//   28 bands              SA:2007-2048
//   amplitude peak/rms    SA:1760-1783

#include <hip/hip_runtime.h>

#include "afx_internal.h"

namespace afx {
namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kThreads = 64 * kWavesPerBlock;
constexpr int kLdsSlots = 1088;  // 1024 complex values + padding of the separable swizzle

template <typename T>
struct cx {
  T re, im;
};

template <typename T>
__device__ __forceinline__ cx<T> cmul(cx<T> a, cx<T> b) {
  return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}

// forward (e^{-i}) radix-4 butterfly, in place: (a,b,c,d) -> (X0,X1,X2,X3)
template <typename T>
__device__ __forceinline__ void radix4(cx<T>& a, cx<T>& b, cx<T>& c, cx<T>& d) {
  const cx<T> t0{a.re + c.re, a.im + c.im}, t1{a.re - c.re, a.im - c.im};
  const cx<T> t2{b.re + d.re, b.im + d.im}, t3{b.re - d.re, b.im - d.im};
  a = {t0.re + t2.re, t0.im + t2.im};
  c = {t0.re - t2.re, t0.im - t2.im};
  b = {t1.re + t3.im, t1.im - t3.re};
  d = {t1.re - t3.im, t1.im + t3.re};
}

// 16-point forward DFT in registers: v[n] -> v[k]
template <typename T>
__device__ __forceinline__ void dft16(cx<T> (&v)[16]) {
  constexpr T c1 = T(0.92387953251128673848), s1 = T(0.38268343236508978178);
  constexpr T rh = T(0.70710678118654752440);
#pragma unroll
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
#pragma unroll
  for (int c = 0; c < 4; ++c) radix4(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
  // v[4c + d] = X[c + 4d]: transpose the 4x4 register grid (pure renaming)
  cx<T> t[16];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int d = 0; d < 4; ++d) t[c + 4 * d] = v[4 * c + d];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = t[i];
}

// LDS view of one wave: complex<float> packs into one 8-byte slot, complex<double> uses two planes
template <typename T>
struct Lds;
template <>
struct Lds<float> {
  float2* p;
  __device__ __forceinline__ void put(int slot, cx<float> v) const { p[slot] = make_float2(v.re, v.im); }
  __device__ __forceinline__ cx<float> get(int slot) const {
    const float2 t = p[slot];
    return {t.x, t.y};
  }
};
template <>
struct Lds<double> {
  double* p;
  __device__ __forceinline__ void put(int slot, cx<double> v) const {
    p[slot] = v.re;
    p[kLdsSlots + slot] = v.im;
  }
  __device__ __forceinline__ cx<double> get(int slot) const { return {p[slot], p[kLdsSlots + slot]}; }
};

// orders this wave's LDS traffic for the compiler; the DS unit executes a wave's ops in order
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// inclusive prefix sum across the 64 lanes
__device__ __forceinline__ double wave_scan_incl(double v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const double t = __shfl_up(v, o);
    if (lane >= o) v += t;
  }
  return v;
}

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

__device__ __forceinline__ double nan_to_zero(double v) { return (v != v) ? 0.0 : v; }

// TAudioMath::LinToDb(double), AudioMath.inl:55-70 (MEpsilon is the float literal 1e-12f)
__device__ __forceinline__ double lin_to_db(double v) {
  if (v == 1.0) return 0.0;
  if (v > (double)1e-12f) return log(v) * 8.685889638065035;  // 20 / ln 10
  return -200.0;
}

template <typename T, typename TIn>
__global__ __launch_bounds__(kThreads, 2) void frames_kernel(const FrameArgs a) {
  using Pair = typename InPair<TIn>::type;
  extern __shared__ __align__(16) unsigned char lds_raw[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  Lds<T> lds;
  lds.p = reinterpret_cast<decltype(lds.p)>(lds_raw + (size_t)wave * kLdsSlots * sizeof(cx<T>));
  double* const lds_mag = reinterpret_cast<double*>(lds_raw + (size_t)wave * kLdsSlots * sizeof(cx<T>));

  // lane coordinates of the three layouts
  const int m2 = lane >> 4, h = (lane >> 2) & 3, q = lane & 3;  // P1 layout: lane = 16 m2 + 4 h + q
  const int lj1 = lane >> 2, lh = lane & 3;                      // P2 layout: lane = 4 j1 + h
  const int e1w = h + 68 * q + 272 * m2;   // + 4 j1
  const int e1r = 4 * lj1 + lh;            // + 68 q + 272 m2
  const int e2w = lj1 + 68 * lh;           // + 16 j2 + 272 q
  const int e2r = lane;                    // + 68 h + 272 q
  const int partner = (64 - lane) & 63;

  const cx<T>* const win = reinterpret_cast<const cx<T>*>(a.win);
  const cx<T>* const t1 = reinterpret_cast<const cx<T>*>(a.t1);
  const cx<T>* const t2 = reinterpret_cast<const cx<T>*>(a.t2);
  const cx<T>* const post = reinterpret_cast<const cx<T>*>(a.post);
  const TIn* const pcm = reinterpret_cast<const TIn*>(a.pcm);

  const int wave_global = blockIdx.x * kWavesPerBlock + wave;
  const int wave_stride = gridDim.x * kWavesPerBlock;

  for (int ci = wave_global; ci < a.n_chunks; ci += wave_stride) {
    const Chunk ch = a.chunks[ci];
    const TIn* src = pcm + ch.sample_off;
    const bool preroll = (ch.flags & kChunkPreroll) != 0;
    const int total = ch.nframes + (preroll ? 1 : 0);

    Pair raw[16];
#pragma unroll
    for (int r = 0; r < 8; ++r) raw[r] = reinterpret_cast<const Pair*>(src)[64 * r + lane];
    double prev_mag[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) prev_mag[r] = 0.0;

    for (int fi = 0; fi < total; ++fi) {
      const TIn* fsrc = src + (size_t)fi * kHop;
#pragma unroll
      for (int r = 8; r < 16; ++r) raw[r] = reinterpret_cast<const Pair*>(fsrc)[64 * r + lane];

      // ---- time-domain descriptors on the hop = rows 0..7 (SA:871-872) ----
      double amp_peak = 0.0, amp_sq = 0.0;
      if (a.mask & ((1u << 11) | (1u << 12))) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const double x0 = (double)raw[r].x, x1 = (double)raw[r].y;
          amp_peak = fmax(amp_peak, fmax(fabs(x0), fabs(x1)));
          amp_sq += x0 * x0 + x1 * x1;
        }
        amp_peak = wave_max(amp_peak);
        amp_sq = wave_sum(amp_sq);
      }

      // ---- window (table already carries the 1/2048 of kDivFwdByN and the 1/2 of the untangle) ----
      cx<T> v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const cx<T> w = win[64 * r + lane];
        v[r] = {(T)raw[r].x * w.re, (T)raw[r].y * w.im};
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) raw[r] = raw[r + 8];

      // ---- P1 + T1 + E1 ----
      dft16(v);
#pragma unroll
      for (int j1 = 1; j1 < 16; ++j1) v[j1] = cmul(v[j1], t1[4 * j1 + m2]);
      wave_lds_fence();
#pragma unroll
      for (int j1 = 0; j1 < 16; ++j1) lds.put(e1w + 4 * j1, v[j1]);
      wave_lds_fence();
#pragma unroll
      for (int mm = 0; mm < 4; ++mm)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) v[4 * mm + qq] = lds.get(e1r + 68 * qq + 272 * mm);

      // ---- P2 + T2 + E2 ----
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) radix4(v[qq], v[4 + qq], v[8 + qq], v[12 + qq]);
#pragma unroll
      for (int g = 0; g < 16; ++g) v[g] = cmul(v[g], t2[64 * g + lane]);
      wave_lds_fence();
#pragma unroll
      for (int j2 = 0; j2 < 4; ++j2)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) lds.put(e2w + 16 * j2 + 272 * qq, v[4 * j2 + qq]);
      wave_lds_fence();
#pragma unroll
      for (int hh = 0; hh < 4; ++hh)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) v[4 * hh + qq] = lds.get(e2r + 68 * hh + 272 * qq);

      // ---- P3: v[k2] = Z[lane + 64 k2] ----
      dft16(v);

      // ---- E3 + untangle + magnitude: mag[r] = |X[lane + 64 r]| ----
      double mag[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        cx<T> p{__shfl(v[15 - r].re, partner), __shfl(v[15 - r].im, partner)};
        if (lane == 0) p = v[(16 - r) & 15];
        const cx<T> z = v[r];
        const cx<T> w = post[64 * r + lane];
        const T er = z.re + p.re, ei = z.im - p.im;   // E = Z + conj(P)
        const T orr = z.im + p.im, oi = p.re - z.re;  // O = -i (Z - conj(P))
        const T xr = er + (w.re * orr - w.im * oi);
        const T xi = ei + (w.re * oi + w.im * orr);
        mag[r] = (double)sqrt(xr * xr + xi * xi);
      }

      const bool emit = !(preroll && fi == 0);
      const int64_t row = (int64_t)ch.frame0 + fi - (preroll ? 1 : 0);
      double* const rec = a.rec + row * a.lay.stride;

      if (emit) {
        if (a.mag_out) {
#pragma unroll
          for (int r = 0; r < 16; ++r) a.mag_out[row * kHalf + 64 * r + lane] = mag[r];
        }

        // ---- MFCC: sparse mel rows, log, 14-point DCT-II (vector.c:350-391) ----
        if (a.mask & 1u) {
          double e[kNumCep];
#pragma unroll
          for (int f = 0; f < kNumCep; ++f) e[f] = 0.0;
#pragma unroll
          for (int r = 0; r < kMelRows; ++r)
#pragma unroll
            for (int f = 0; f < kNumCep; ++f)
              if (mel_touches(f, r)) e[f] += mag[r] * a.melw[64 * mel_pair_index(r, f) + lane];
          double mine = 0.0;  // lane f keeps log e[f]
#pragma unroll
          for (int f = 0; f < kNumCep; ++f) {
            const double s = wave_sum(e[f]);
            if ((lane & 15) == f) mine = s;
          }
          mine = log(mine < 2e-42 ? 2e-42 : mine);
          double c = 0.0;
          const int n = lane & 15;
#pragma unroll
          for (int m = 0; m < kNumCep; ++m) c += __shfl(mine, m) * a.dct[16 * (n < kNumCep ? n : 0) + m];
          if (lane < kNumCep) rec[a.lay.mfcc + lane] = c;
        }

        // ---- spectral statistics over bins 1..738, j = bin - 1 (SA:1808-1933) ----
        if (a.mask & 0x1FEu) {
          double s1 = 0.0, s2 = 0.0, sj = 0.0, prod = 1.0;
          double fa = 0.0, fb = 0.0, faa = 0.0, fbb = 0.0, fab = 0.0;
          const bool first = (fi == 0) && !preroll;  // SA:937-940: frame 0 is compared with itself
#pragma unroll
          for (int r = 0; r < 12; ++r) {
            const int k = 64 * r + lane;
            const bool ok = (r == 0) ? (lane >= kFirstBin) : (r == 11 ? (k <= kLastBin) : true);
            const double m = ok ? mag[r] : 0.0;
            s1 += m;
            s2 += m * m;
            sj += (double)(k - kFirstBin) * m;
            prod *= ok ? (m + 1e-20) : 1.0;
            const double b = first ? m : (ok ? prev_mag[r] : 0.0);
            fa += m; fb += b; faa += m * m; fbb += b * b; fab += m * b;
          }
          s1 = wave_sum(s1);
          const double n = (double)kBinCount;
          if (a.mask & (1u << 1)) {
            s2 = wave_sum(s2);
            if (lane == 0) rec[a.lay.srms] = nan_to_zero(sqrt(s2 / n));
          }
          double cen = 0.0, spr = 0.0;
          if (a.mask & 0x3Cu) {  // centroid, spread, skewness, kurtosis
            sj = wave_sum(sj);
            if (s1 != 0.0) {
              cen = sj / s1;
              double sv = 0.0;
#pragma unroll
              for (int r = 0; r < 12; ++r) {
                const int k = 64 * r + lane;
                const bool ok = (r == 0) ? (lane >= kFirstBin) : (r == 11 ? (k <= kLastBin) : true);
                const double t = (double)(k - kFirstBin) - cen;
                sv += ok ? t * t * mag[r] : 0.0;
              }
              spr = wave_sum(sv) / s1;
            }
            if (lane == 0) {
              if (a.lay.centroid >= 0) rec[a.lay.centroid] = cen;
              if (a.lay.spread >= 0) rec[a.lay.spread] = spr;
            }
            if (a.mask & 0x30u) {
              double sk = 0.0, ku = 0.0;
              if (fabs(spr) > (double)1e-12f) {
#pragma unroll
                for (int r = 0; r < 12; ++r) {
                  const int k = 64 * r + lane;
                  const bool ok = (r == 0) ? (lane >= kFirstBin) : (r == 11 ? (k <= kLastBin) : true);
                  const double t = (mag[r] - cen) / spr;
                  const double tt = t * t;
                  sk += ok ? tt * t : 0.0;
                  ku += ok ? tt * tt : 0.0;
                }
                sk = wave_sum(sk) / n;
                ku = wave_sum(ku) / n - 3.0;
              }
              if (lane == 0) {
                if (a.lay.skew >= 0) rec[a.lay.skew] = sk;
                if (a.lay.kurt >= 0) rec[a.lay.kurt] = ku;
              }
            }
          }
          if (a.mask & (1u << 7)) {  // flatness: GM / AM in dB / -60, clamped (SA:129-133)
            const double sumlog = wave_sum(log(prod));
            const double gm = exp(sumlog / n);
            const double am = s1 / n;
            const double fl = (am == 0.0) ? 0.0 : gm / am;
            const double d = lin_to_db(fl) / -60.0;
            if (lane == 0) rec[a.lay.flatness] = nan_to_zero(d < 1.0 ? d : 1.0);
          }
          if (a.mask & (1u << 8)) {  // flux = Pearson correlation with the previous frame
            fa = wave_sum(fa); fb = wave_sum(fb); faa = wave_sum(faa); fbb = wave_sum(fbb);
            fab = wave_sum(fab);
            const double ma = fa / n, mb = fb / n;
            const double denom2 = (faa - ma * ma * n) * (fbb - mb * mb * n);
            const double num = fab - (ma * mb * n);
            const double fx = (fabs(denom2) > (double)1e-12f) ? num / sqrt(denom2) : 0.0;
            if (lane == 0) rec[a.lay.flux] = fx;
          }
          if (a.mask & (1u << 6)) {  // rolloff (scalar.c:472-492): 43 * #elements until 85 %
            // natural-order copy in LDS, then each lane walks 12 consecutive bins
            wave_lds_fence();
#pragma unroll
            for (int r = 0; r < 12; ++r) lds_mag[64 * r + lane] = mag[r];
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
        if (a.mask & (1u << 9)) {
          double mine = 0.0;
#pragma unroll
          for (int b = 0; b < kNumBands; ++b) {
            double acc = 0.0;
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (band_touches(b, r)) {
                const int k = 64 * r + lane;
                acc += (k >= kBandEdge[b] && k < kBandEdge[b + 1]) ? mag[r] * mag[r] : 0.0;
              }
            acc = wave_sum(acc);
            if (lane == b) mine = acc;
          }
          if (lane < kNumBands) rec[a.lay.bands + lane] = mine;
        }

        if (lane == 0) {
          if (a.mask & (1u << 11)) rec[a.lay.amp_peak] = amp_peak;
          if (a.mask & (1u << 12)) rec[a.lay.amp_rms] = nan_to_zero(sqrt(amp_sq / (double)kHop));
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) prev_mag[r] = mag[r];
    }
  }
}

}  // namespace

int frames_block_threads() { return kThreads; }
int frames_lds_bytes(int precision) {
  return kWavesPerBlock * kLdsSlots * (precision == 0 ? (int)sizeof(cx<double>) : (int)sizeof(cx<float>));
}

hipError_t launch_frames(const FrameArgs& a, int precision, int pcm_dtype, int grid_blocks,
                         hipStream_t stream) {
  if (a.n_chunks <= 0) return hipSuccess;
  const dim3 grid(grid_blocks), block(kThreads);
  const size_t lds = (size_t)frames_lds_bytes(precision);
  if (precision == 0) {
    if (pcm_dtype == 0)
      hipLaunchKernelGGL((frames_kernel<double, float>), grid, block, lds, stream, a);
    else
      hipLaunchKernelGGL((frames_kernel<double, double>), grid, block, lds, stream, a);
  } else {
    if (pcm_dtype == 0)
      hipLaunchKernelGGL((frames_kernel<float, float>), grid, block, lds, stream, a);
    else
      hipLaunchKernelGGL((frames_kernel<float, double>), grid, block, lds, stream, a);
  }
  return hipGetLastError();
}

hipError_t launch_bands(const BandArgs&, hipStream_t) { return hipErrorNotSupported; }

}  // namespace afx
