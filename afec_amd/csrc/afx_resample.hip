// afec_amd/csrc/afx_resample.hip -- the sample-rate conversion of LoadSample on the GPU.
//
// TSampleAnalyser::LoadSample converts files that are not at the analyser's rate with libresample 0.1.3
// (SampleAnalyser.cpp:563-607: resample_open(1, f, f); resample_process(h, f, in, n, 1, &used, out, NewSize), f = 1 /
// Speed), on the mono mix, before rms / peak / normalisation / trim.  Three kernels between the upload of a batch's
// decoded PCM and the scan kernels of afx_load.hip:
//
//   resample_mix_kernel     decoded PCM -> the mono "16-bit float" buffer (mono_sample of afx_load.hip)
//   resample_plan_kernel    one lane per file walks the converter's clock: libresample advances its time by
//                           CurrentTime += dt in double (resamplesubs.c:49-62, 104-116), window after window of
//                           4096 input samples with the "Time -= Nx" / creep bookkeeping of resample_process
//                           (resample.c:240-292).  Every sum is rounded, so the clock is a recurrence, not k * dt:
//                           it is walked serially, and for every 16 output samples the lane leaves a record {time of
//                           the first sample, input offset of the window} -- and, when a new window starts inside
//                           the 16, where, with the time and offset from there on.
//   resample_filter_kernel  an output sample steps the clock up to 15 times from its record, then adds the two wings
//                           of the Kaiser-windowed sinc around its input position in the library's order
//                           (filterkit.c:115-215: lrsFilterUp for factor >= 1, lrsFilterUD below; float products
//                           and float sums, no contraction; interpFilt is FALSE, resample.c:171).
//
// The library's input window X is only bookkeeping here (where it starts in the input): every tap reads the mono buffer
// directly and samples outside [0, n) are the zeros the library pads with (Xoff in front, Xoff behind the last sample).
// Bit-exact against the oracle (tests/test_gpu_resample.py), which is bit-exact against the reference's libresample.
//
// Cost: 2 x ~18 taps per output sample, each a pick from the 272 KiB coefficient table and a load of the input
// (neighbouring lanes read neighbouring samples); the picks go through LDS (see resample_filter_kernel).
#include <hip/hip_runtime.h>

#include "afx_internal.h"

// No contraction anywhere in this file: the reference's products and sums are separate roundings.  (Plain operators, not
// __fmul_rn / __fadd_rn: those are header functions compiled under the default, contractible, and fuse once inlined.)
#pragma clang fp contract(off)

namespace afx {
namespace {

constexpr int kNpc = 4096;                          // resample_defs.h:72
constexpr int kNwing = kNpc * (35 - 1) / 2;         // Nmult = 35 (highQuality), resample.c:104-108
constexpr int kGroup = 16;                          // output samples per group record

// the reference's decoders hand LoadSample "16-bit floats": the same conversions as afx_load.hip
__device__ __forceinline__ float rs_to_16bit_float(const unsigned char* raw, int format, int64_t idx) {
  if (format == 0) return (float)reinterpret_cast<const short*>(raw)[idx];
  if (format == 1) {
    const unsigned char* b = raw + 3 * idx;
    const int v = (int)(((unsigned)b[0] | ((unsigned)b[1] << 8) | ((unsigned)b[2] << 16)) << 8);
    return (float)((double)v * 32768.0 / 2147483648.0);
  }
  if (format == 3) {
    const float v = (float)((double)reinterpret_cast<const int*>(raw)[idx] * 32768.0 / 2147483648.0);
    return fmaxf(-32768.0f, fminf(32767.0f, v));
  }
  const double d = (format == 4 ? reinterpret_cast<const double*>(raw)[idx] : (double)reinterpret_cast<const float*>(raw)[idx]) * 32768.0;
  return (float)(d < -32768.0 ? -32768.0 : (d > 32767.0 ? 32767.0 : d));
}

__global__ __launch_bounds__(256) void resample_mix_kernel(unsigned char* raw, const ResampleFile* files) {
  const ResampleFile f = files[blockIdx.x];
  const unsigned char* src = raw + f.raw_off;
  float* mono = reinterpret_cast<float*>(raw + f.mono_off) + kResampleMargin;
  // the zeros libresample pads its input with (Xoff in front of the first and behind the last sample; the filter
  // kernel reads them instead of testing bounds)
  if (blockIdx.y == 0)
    for (int k = threadIdx.x; k < kResampleMargin; k += 256) { mono[k - kResampleMargin] = 0.0f; mono[f.n_in + k] = 0.0f; }
  for (int64_t n = (int64_t)blockIdx.y * 256 + threadIdx.x; n < f.n_in; n += (int64_t)gridDim.y * 256) {
    float d = rs_to_16bit_float(src, f.format, n * f.channels);      // SampleAnalyser.cpp:535-548
    if (f.channels > 1) {
      for (int c = 1; c < f.channels; ++c) d = d + rs_to_16bit_float(src, f.format, n * f.channels + c);
      d = d * (1.0f / (float)f.channels);
    }
    mono[n] = d;
  }
}

__global__ __launch_bounds__(64) void resample_plan_kernel(unsigned char* raw, const ResampleFile* files, int n_files,
                                                           ResampleGroup* groups) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n_files) return;
  const ResampleFile f = files[i];
  ResampleGroup* recs = groups + f.group_off;
  const double factor = f.factor;
  const double dt = 1.0 / factor;                                                      // resamplesubs.c:45
  const double reach = ((35 + 1) / 2.0) * fmax(1.0, 1.0 / factor) + 10;               // resample.c:133-135
  const unsigned xoff = (unsigned)reach;
  const unsigned xsize = (2 * xoff + 10 > 4096u) ? 2 * xoff + 10 : 4096u;             // resample.c:142
  unsigned xread = xoff;
  double time = (double)xoff;
  int64_t base = -(int64_t)xoff, used = 0;
  int64_t written = 0;
  for (;;) {
    int64_t len = (int64_t)xsize - xread;
    if (len >= f.n_in - used) len = f.n_in - used;
    used += len;
    xread += (unsigned)len;
    const int nx = (used == f.n_in) ? (int)xread - (int)xoff : (int)xread - 2 * (int)xoff;   // resample.c:240-250
    if (nx <= 0) break;
    double t = time;
    const double end_time = t + (double)nx;
    // a window that starts inside a record (not at a multiple of kGroup outputs): the record's second part
    if (t < end_time && written < f.n_out && (written & (kGroup - 1)) != 0) {
      ResampleGroup& g = recs[written >> 4];
      g.cut = (int32_t)(written & (kGroup - 1)); g.t1 = t; g.base1 = (int32_t)base;
    }
    const double safe_end = end_time - 17.0 * dt;   // sixteen more steps from below this stay inside the window
    while (t < end_time) {                                                             // resamplesubs.c:49-62
      if (written < f.n_out && (written & (kGroup - 1)) == 0) {
        ResampleGroup g;
        g.t0 = t; g.t1 = 0.0; g.base0 = (int32_t)base; g.base1 = 0; g.cut = kGroup; g.pad = 0;
        recs[written >> 4] = g;
        if (t < safe_end) {                         // a whole record inside the window: the same sixteen additions, no tests
#pragma unroll
          for (int q = 0; q < kGroup; ++q) t += dt;
          written += kGroup;
          continue;
        }
      }
      ++written;
      t += dt;
    }
    time = t;
    time -= (double)nx;                                                                // resample.c:271-280
    unsigned xp = xoff + (unsigned)nx;
    const int ncreep = (int)time - (int)xoff;
    if (ncreep) { time -= (double)ncreep; xp += (unsigned)ncreep; }
    base += (int64_t)xp - (int64_t)xoff;                                               // resample.c:283-292: the window moves on
    xread = xread - (xp - xoff);
    if (written >= f.n_out) break;                                                     // resample.c:305-318: the output buffer is full
  }
  // what the converter did not produce (the reference leaves it uninitialised and asserts that there is none, SA:596-597):
  // zeros, and records that say so (count of produced samples in `pad` of the file's first record is not needed: a record
  // that was never started keeps cut = 0 from the clear below)
  float* out = reinterpret_cast<float*>(raw + f.out_off);
  for (int64_t k = written; k < f.n_out; ++k) out[k] = 0.0f;
  const int64_t n_recs = (f.n_out + kGroup - 1) / kGroup;
  for (int64_t r = (written + kGroup - 1) / kGroup; r < n_recs; ++r) {
    ResampleGroup g;
    g.t0 = 0.0; g.t1 = 0.0; g.base0 = 0; g.base1 = 0; g.cut = 0; g.pad = 1;   // pad = 1: nothing was produced here
    recs[r] = g;
  }
  if (written < f.n_out && (written & (kGroup - 1)) != 0) recs[written >> 4].pad = 2 + (int32_t)(written & (kGroup - 1));   // produced: the first (pad - 2)
}

// One workgroup: kRsThreads x kRsPer consecutive output samples of one file (thread `tid` owns outputs tid, tid +
// kRsThreads, ..: neighbouring lanes read neighbouring input samples).  The taps are the OUTER loop: tap i of either wing
// reads the coefficient table inside [i dh, (i + 1) dh] (dh = 4096 min(1, factor) table steps per input sample), so
// that part of the table is staged in LDS once per tap and every output sample of the workgroup picks its coefficient
// from there -- lanes have unrelated phases, and the same picks from global memory are 64 different cache lines per
// load instruction (62.7 ms for 12 500 one-second 48 kHz files that way, profiles/r03/README.md).  Every output still adds its
// taps in the library's order: left wing i = 0, 1, .. into v, right wing into w, then (v + w) LpScl.
// Shape of a workgroup, measured on MI355X for 12 500 one-second files (48 kHz / 96 kHz / 22.05 kHz -> 44.1 kHz; filter
// kernel only): 1024 x 4 outputs at 4 waves per SIMD 13.7 / 24.3 / 10.6 ms; 512 x 8 15.2 / 26.4 / 12.4; 256 x 8 16.2 /
// 26.7 / 13.0; 256 x 16 (spills) 58; one 16-lane group per record picking from global memory 62.7 (HISTORY.md, round 3).
constexpr int kRsThreads = 1024, kRsPer = 4, kRsOcc = 4, kRsHalf = kRsPer / 2, kRsBlockOut = kRsThreads * kRsPer, kRsWindow = 4112;

__global__ __launch_bounds__(kRsThreads, kRsOcc) void resample_filter_kernel(unsigned char* raw, const ResampleFile* files, int n_files,
                                                                     const ResampleGroup* groups, const float* __restrict__ imp) {
  __shared__ __align__(16) float s_win[kRsWindow];
  const int tid = threadIdx.x;
  // the file of this workgroup: the last one whose first workgroup is <= blockIdx.x
  int lo = 0, hi = n_files - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (files[mid].block_off <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const ResampleFile f = files[lo];
  const int64_t o_base = ((int64_t)blockIdx.x - f.block_off) * kRsBlockOut;
  const ResampleGroup* recs = groups + f.group_off;
  const float* mono0 = reinterpret_cast<const float*>(raw + f.mono_off);   // kResampleMargin zeros, the samples, kResampleMargin zeros
  float* out = reinterpret_cast<float*>(raw + f.out_off);
  const double factor = f.factor;
  const double dt = 1.0 / factor;
  const bool up = factor >= 1;
  const double dh = up ? (double)kNpc : fmin((double)kNpc, factor * kNpc);   // resamplesubs.c:91

  // per output sample: the converter's time -> centre input sample and the two wings' positions in the table
  double ho_l[kRsPer], ho_r[kRsPer];   // lrsFilterUD: Ho of the next tap; lrsFilterUp: the integer index (exact in a double)
  unsigned cc[kRsPer];   // centre input sample + kResampleMargin: an unsigned offset from the buffer's start
  float v[kRsPer], w[kRsPer];
  unsigned live = 0;   // bit j: output j of this thread exists
#pragma unroll
  for (int j = 0; j < kRsPer; ++j) {
    const int64_t o = o_base + (int64_t)j * kRsThreads + tid;
    v[j] = 0.0f; w[j] = 0.0f; cc[j] = kResampleMargin; ho_l[j] = ho_r[j] = (double)kNwing;   // beyond the table: no tap
    if (o < f.n_out) {
      const ResampleGroup g = recs[o >> 4];
      const int k = (int)(o & (kGroup - 1));
      const bool produced = (g.pad == 0) || (g.pad >= 2 && k < g.pad - 2);
      if (produced) {
        const bool second = k >= g.cut;
        double t = second ? g.t1 : g.t0;
        const int steps = second ? k - g.cut : k;
        for (int q = 0; q < steps; ++q) t += dt;                   // the converter's clock, step by step
        const double left_phase = t - floor(t), right_phase = 1.0 - left_phase;
        cc[j] = (unsigned)((second ? g.base1 : g.base0) + (int)t + kResampleMargin);
        live |= 1u << j;
        if (up) {                                                   // filterkit.c:115-145
          ho_l[j] = (double)(int)(left_phase * kNpc);
          const double ph = right_phase * kNpc;
          ho_r[j] = (double)((int)ph + (ph == 0 ? kNpc : 0));
        } else {                                                    // filterkit.c:172-196
          ho_l[j] = left_phase * dh;
          ho_r[j] = right_phase * dh;
          if (right_phase == 0) ho_r[j] += dh;
        }
      }
    }
  }
  const int n_taps = (int)((double)kNwing / dh) + 2;
  for (int i = 0; i < n_taps; ++i) {
    // the part of the table tap i can touch: Ho = (phase + i) dh up to the rounding of its i additions (far below one
    // table step), phase in [0, 1] -> two entries of slack on either side
    int w_lo = (int)((double)i * dh) - 2;
    int w_hi = (int)((double)(i + 1) * dh) + 3;
    w_lo = (w_lo < 0 ? 0 : w_lo) & ~3;                              // 16-byte steps: the window is staged four values at a time
    w_lo = w_lo > kNwing - 4 ? kNwing - 4 : w_lo;                   // (the last, spare tap lies behind the table: nothing picks from it)
    w_hi = w_hi > kNwing ? kNwing : w_hi;
    const int w_last = w_hi - w_lo - 1;
    __syncthreads();
    for (int e = 4 * tid; e <= w_last; e += 4 * kRsThreads)         // (kNwing is a multiple of 4: the last four never pass the table's end)
      *reinterpret_cast<float4*>(s_win + e) = *reinterpret_cast<const float4*>(imp + w_lo + e);
    __syncthreads();
    // no branches: every pick and every input sample is loaded (clamped picks of finished wings are not used; the input
    // has kResampleMargin zeros on either side), then the products are added where the wing still runs; kRsHalf output
    // samples at a time
#pragma unroll
    for (int j0 = 0; j0 < kRsPer; j0 += kRsHalf) {
      float cl[kRsHalf], cr[kRsHalf], xl[kRsHalf], xr[kRsHalf];
      int hl[kRsHalf], hr[kRsHalf];
#pragma unroll
      for (int q = 0; q < kRsHalf; ++q) {
        const int j = j0 + q;
        hl[q] = (int)ho_l[j];
        hr[q] = (int)ho_r[j];
        const int rl = min(max(hl[q] - w_lo, 0), w_last), rr = min(max(hr[q] - w_lo, 0), w_last);
        cl[q] = s_win[rl];
        cr[q] = s_win[rr];
        xl[q] = mono0[cc[j] - (unsigned)i];
        xr[q] = mono0[cc[j] + 1u + (unsigned)i];
      }
#pragma unroll
      for (int q = 0; q < kRsHalf; ++q) {
        const int j = j0 + q;
        const float pl = cl[q] * xl[q], pr = cr[q] * xr[q];
        const float vn = v[j] + pl, wn = w[j] + pr;
        v[j] = (hl[q] < kNwing) ? vn : v[j];                        // left wing: End = &Imp[Nwing]
        w[j] = (hr[q] < kNwing - 1) ? wn : w[j];                    // right wing: End = &Imp[Nwing] - 1
        ho_l[j] += dh;                                              // Ho += dhb / Hp += Npc (exact for the integer steps of lrsFilterUp)
        ho_r[j] += dh;
      }
    }
  }
  float lpscl = 1.0f;
  if (factor < 1) lpscl = (float)((double)lpscl * factor);          // resample.c:211-212
#pragma unroll
  for (int j = 0; j < kRsPer; ++j) {
    const int64_t o = o_base + (int64_t)j * kRsThreads + tid;
    if (live & (1u << j)) {
      float r = v[j] + w[j];                                        // resamplesubs.c:54-58
      r = r * lpscl;
      out[o] = r;
    }
  }
}

}  // namespace

int64_t resample_blocks(int64_t n_out) { return (n_out + kRsBlockOut - 1) / kRsBlockOut; }

hipError_t launch_resample(unsigned char* raw, const ResampleFile* files, int n_files, int64_t n_blocks, int64_t max_n_in,
                           ResampleGroup* groups, const float* imp, hipStream_t stream) {
  if (n_files <= 0) return hipSuccess;
  const int per_file = (int)std::min<int64_t>(std::max<int64_t>((max_n_in + 256 * 8 - 1) / (256 * 8), 1), n_files >= 256 ? 4 : 256);
  resample_mix_kernel<<<dim3((unsigned)n_files, (unsigned)per_file), 256, 0, stream>>>(raw, files);
  resample_plan_kernel<<<(n_files + 63) / 64, 64, 0, stream>>>(raw, files, n_files, groups);
  if (n_blocks > 0)
    resample_filter_kernel<<<(unsigned)n_blocks, kRsThreads, 0, stream>>>(raw, files, n_files, groups, imp);
  return hipGetLastError();
}

}  // namespace afx
