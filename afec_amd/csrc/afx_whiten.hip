// afec_amd/csrc/afx_whiten.hip -- whitened-spectrum neighbours of the spectral loop, gfx950.
//
// The follower of the adaptive whitening is the only state that crosses frames (SampleAnalyser.cpp (SA)
// 805-809, 849-858).  follow_kernel runs that recurrence alone -- one wave per (buffer, 64 bins), frames in
// order -- and leaves the follower at every chunk start; whiten_kernel then takes one chunk of consecutive
// frames per wave (its own chunk table: in batches of thousands of short files a chunk is a whole file, which starts
// from the reset state and needs no follow_kernel at all) and does, per frame:
//
//   adaptive whitening   aubio_spectral_whitening_do, Aubio spectral/awhitening.c:41-51
//   peak spectrum        SCreatePeakSpectrum, SA:95-123 -> TStatistics::Peaks, Statistics.cpp:140-232
//   spectral_complexity  non-zero bins of the peak spectrum inside the analysis range, SA:1937-1947
//   failsafe_f0          SA:897-916: yinfast's f0 when confident, else the spectral centroid (audible frames)
//   spectral_inharmonicity, tristimulus1..3: LibXtract reads the partial frequencies from the upper half
//     of the fft-sized buffer (Xt/src/scalar.c:304-326, 638-661, vector.c:540-579), which SA:844-845 clears,
//     so every partial sits at 0 Hz and all four are 0 for every frame (SURVEY 8a note; checked against the
//     reference's objects in tests/golden/neighbours.npz)
//
// Bins are held 16 per lane in order (bin = 16 lane + i), so the peak scan needs its neighbours from
// the next / previous lane only at the two ends.

#include <hip/hip_runtime.h>

#include "../../include/afx.h"
#include "afx_internal.h"
#include "afx_device.h"

namespace afx {
namespace {

__device__ __forceinline__ void load_bins(const double* p, double (&m)[16]) {
  const double2* q = reinterpret_cast<const double2*>(p);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const double2 v = q[j];
    m[2 * j] = v.x; m[2 * j + 1] = v.y;
  }
}

// follower recurrence only: one wave per (buffer, 64 bins), frames in order, state saved at chunk starts
__global__ __launch_bounds__(256) void follow_kernel(const WhitenArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t wave_global = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t wave_stride = (int64_t)gridDim.x * 4;
  constexpr int kAhead = 32;
  for (int64_t w = wave_global; w < (int64_t)a.n_bufs * 16; w += wave_stride) {
    const int b = (int)(w >> 4), k = 64 * (int)(w & 15) + lane;
    if (a.chunk_first[b + 1] - a.chunk_first[b] <= 1) continue;   // a buffer that is one chunk starts from the reset state
    const int64_t f_begin = a.frame_offset[b], n = a.frame_offset[b + 1] - f_begin;
    const double* const col = a.mag + f_begin * kHalf + k;
    double* const save = a.follower + (int64_t)a.chunk_first[b] * kHalf + k;
    double follow = a.floor_value;                              // aubio_spectral_whitening_reset
    // frames in batches of kAhead loads, whatever the chunk length (a lone 20 s file is cut into one-frame
    // chunks to fill the chip: the loads must not wait for chunk boundaries)
    const int n32 = (int)n, per_chunk = a.chunk_frames;
    int next_save = 0, ci = 0;
    for (int f0 = 0; f0 < n32; f0 += kAhead) {
      double m[kAhead];
#pragma unroll
      for (int j = 0; j < kAhead; ++j) m[j] = (f0 + j < n32) ? col[(int64_t)(f0 + j) * kHalf] : 0.0;
#pragma unroll
      for (int j = 0; j < kAhead; ++j)
        if (f0 + j < n32) {
          if (f0 + j == next_save) {                             // state at a chunk's first frame
            save[(int64_t)ci * kHalf] = follow;
            ++ci;
            next_save += per_chunk;
          }
          double t = a.decay * follow;
          t = (t > a.floor_value) ? t : a.floor_value;
          follow = (m[j] > t) ? m[j] : t;
        }
    }
  }
}

__global__ __launch_bounds__(256) void whiten_kernel(const WhitenArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int wave_stride = gridDim.x * 4;
  const bool want_cplx = (a.mask & AFX_D_SPECTRAL_COMPLEXITY) != 0;
  const bool want_f0 = (a.mask & AFX_D_F0) != 0;

  for (int ci = wave_global; ci < a.n_chunks; ci = next_item(a.queue, ci, min(wave_stride, a.n_chunks), wave_stride, lane)) {
    const Chunk ch = a.chunks[ci];
    const int64_t f_begin = ch.frame0, f_end = f_begin + ch.nframes;
    double follow[16];
    if (want_cplx) {
      if (ch.flags & kChunkFirstOfBuffer) {
#pragma unroll
        for (int i = 0; i < 16; ++i) follow[i] = a.floor_value;     // aubio_spectral_whitening_reset
      } else load_bins(a.follower + (int64_t)ci * kHalf + 16 * lane, follow);
    }
    double m[16];
    if ((want_cplx || want_f0) && f_begin < f_end) load_bins(a.mag + f_begin * kHalf + 16 * lane, m);
    for (int64_t f = f_begin; f < f_end; ++f) {
      double* const rec = a.rec + f * a.lay.stride;
      double cur[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) cur[i] = m[i];
      if ((want_cplx || want_f0) && f + 1 < f_end) load_bins(a.mag + (f + 1) * kHalf + 16 * lane, m);

      if (want_cplx) {
        double w[16], top = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          double t = a.decay * follow[i];
          t = (t > a.floor_value) ? t : a.floor_value;           // MAX(r_decay * peak, floor)
          follow[i] = (cur[i] > t) ? cur[i] : t;                 // MAX(norm, tmp)
          // (fast_div: exact when the quotient is representable -- 1.0 where the follower sits on the magnitude, 0 for a
          // zero magnitude, the cases that make runs of equal values --, within an ulp otherwise; the magnitudes themselves
          // differ from the reference's in their last bits, so a correctly rounded quotient of them is no closer to its;
          // the generic division is 35 instructions against 8, sixteen times per lane and frame)
          w[i] = fast_div(cur[i], follow[i]);
          top = fmax(top, w[i]);
        }
        top = wave_max(top);
        const double thr = 0.25 * top;                           // MPeakThreshold, SA:47, 105-106
        const double before = prev_lane(w[15], w[15]), after = next_lane(w[0], w[0]);
        // Statistics.cpp:140-232 as a run analysis: a peak is a maximal run of equal values [s..e] entered by a
        // strict rise (w[s-1] < w[s], s >= 1) and left by a strict fall (w[e+1] < w[e], e + 1 < N - 1), above
        // the threshold; it is reported at bin (s + e) / 2.  The two boundary bins and bin N-2 have their own
        // rules there but lie outside the analysis range counted here.
        int start[16], open = -1;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int k = 16 * lane + i;
          const double lo = (i == 0) ? before : w[i - 1];
          const bool begins = (k == 0) || (lo != w[i]);
          const bool rises = (k > 0) && (lo < w[i]);
          if (begins) open = (k << 1) | (rises ? 1 : 0);
          start[i] = open;
        }
        const int incl = wave_scan_max_i(open);
        const int carried = prev_lane(incl, -1);
        int count = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int k = 16 * lane + i;
          const double hi = (i == 15) ? after : w[i + 1];
          const int s = (start[i] >= 0) ? start[i] : carried;
          const bool falls = (k + 1 < kHalf - 1) && (hi < w[i]);
          const int bin = ((s >> 1) + k) >> 1;
          if (falls && (s & 1) && w[i] > thr && bin >= kFirstBin && bin < kFirstBin + kBinCount) ++count;
        }
        count = wave_sum_i(count);
        if (lane == 0) rec[a.lay.complexity] = (double)count;
      }
      if (want_f0) {
        // TStatistics::Centroid over the 1024 magnitudes (Statistics.cpp:459-477)
        double s = 0.0, sk = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          s += cur[i];
          sk += (double)(16 * lane + i) * cur[i];
        }
        s = wave_sum(s);
        sk = wave_sum(sk);
        if (lane == 0) {
          const double f0 = rec[a.lay.f0], conf = rec[a.lay.f0_conf];
          const bool silent = rec[a.lay.f0_safe] != 0.0;         // parked there by pitch_kernel
          double safe = 0.0;
          if (f0 > 0.0 && conf > 0.2) safe = f0;                 // MLowPitchConfidenceValue, SA:64, 897-901
          else if (!silent) {
            const double cb = (s == 0.0) ? 0.0 : sk / s;
            safe = (double)kSampleRate / (double)kFft * (cb > 0.0 ? cb : 0.0);
          }
          rec[a.lay.f0_safe] = safe;
        }
      }
      if (lane == 0) {
        if (a.lay.inharm >= 0) rec[a.lay.inharm] = 0.0;
        if (a.lay.tri1 >= 0) { rec[a.lay.tri1] = 0.0; rec[a.lay.tri2] = 0.0; rec[a.lay.tri3] = 0.0; }
      }
    }
  }
}

}  // namespace

hipError_t launch_whiten(const WhitenArgs& a, hipStream_t stream) {
  if (a.n_chunks <= 0) return hipSuccess;
  if ((a.mask & AFX_D_SPECTRAL_COMPLEXITY) && a.need_follow) {
    const int64_t want = ((int64_t)a.n_bufs * 16 + 3) / 4;
    hipLaunchKernelGGL(follow_kernel, dim3((unsigned)(want < 8192 ? want : 8192)), dim3(256), 0, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  const int want = (a.n_chunks + 3) / 4;
  static const int resident = resident_blocks(whiten_kernel, 256, 0);
  const int cap = a.queue.counter ? resident : 8192;
  hipLaunchKernelGGL(whiten_kernel, dim3(want < cap ? want : cap), dim3(256), 0, stream, a);
  return hipGetLastError();
}

}  // namespace afx
