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
//                           it is walked serially, and every 16 output samples (and at every window start) the lane
//                           leaves a group record {time of the first sample, input offset of the window, first output}.
//   resample_filter_kernel  one 16-lane group per record: lane k steps the clock k times from the record, then adds
//                           the two wings of the Kaiser-windowed sinc around its input position in the library's
//                           order (filterkit.c:115-215: lrsFilterUp for factor >= 1, lrsFilterUD below; float
//                           products and float sums, no contraction; interpFilt is FALSE, resample.c:171).
//
// The library's input window X is only bookkeeping here (where it starts in the input): every tap reads the mono buffer
// directly and samples outside [0, n) are the zeros the library pads with (Xoff in front, Xoff behind the last sample).
// Bit-exact against the oracle (tests/test_gpu_resample.py), which is bit-exact against the reference's libresample.
//
// Cost: 2 x ~18 taps per output sample, each a gather from the 272 KiB coefficient table (L2) and a load of the
// input (neighbouring lanes read neighbouring samples): bound by the gathers, not by HBM.
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
  float* mono = reinterpret_cast<float*>(raw + f.mono_off);
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
                                                           ResampleGroup* groups, int32_t* n_groups) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n_files) return;
  const ResampleFile f = files[i];
  ResampleGroup* out_groups = groups + f.group_off;
  const double factor = f.factor;
  const double dt = 1.0 / factor;                                                      // resamplesubs.c:45
  const double reach = ((35 + 1) / 2.0) * fmax(1.0, 1.0 / factor) + 10;               // resample.c:133-135
  const unsigned xoff = (unsigned)reach;
  const unsigned xsize = (2 * xoff + 10 > 4096u) ? 2 * xoff + 10 : 4096u;             // resample.c:142
  unsigned xread = xoff;
  double time = (double)xoff;
  int64_t base = -(int64_t)xoff, used = 0;
  int64_t written = 0;
  int ng = 0;
  for (;;) {
    int64_t len = (int64_t)xsize - xread;
    if (len >= f.n_in - used) len = f.n_in - used;
    used += len;
    xread += (unsigned)len;
    const int nx = (used == f.n_in) ? (int)xread - (int)xoff : (int)xread - 2 * (int)xoff;   // resample.c:240-250
    if (nx <= 0) break;
    double t = time;
    const double end_time = t + (double)nx;
    while (t < end_time) {                                                             // resamplesubs.c:49-62
      // a group: up to kGroup outputs of this window, the first one at time t
      const double t0 = t;
      int count = 0;
      do {
        ++count;
        t += dt;
      } while (count < kGroup && t < end_time);
      if (written < f.n_out && ng < f.group_cap) {
        const int64_t room = f.n_out - written;
        ResampleGroup g;
        g.t0 = t0; g.base = (int32_t)base; g.out0 = (int32_t)written; g.count = (int32_t)(count < room ? count : room); g.pad = 0;
        out_groups[ng++] = g;
      }
      written += count;
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
  n_groups[i] = ng;
  // what the converter did not produce (the reference leaves it uninitialised and asserts that there is none, SA:596-597)
  float* out = reinterpret_cast<float*>(raw + f.out_off);
  for (int64_t k = written; k < f.n_out; ++k) out[k] = 0.0f;
}

__device__ __forceinline__ float rs_at(const float* in, int64_t n, int64_t i) { return (i >= 0 && i < n) ? in[i] : 0.0f; }

__global__ __launch_bounds__(256) void resample_filter_kernel(unsigned char* raw, const ResampleFile* files, int n_files,
                                                              const ResampleGroup* groups, const int32_t* n_groups,
                                                              const float* __restrict__ imp) {
  const int64_t slot = (int64_t)blockIdx.x * (256 / kGroup) + (threadIdx.x / kGroup);
  const int k = threadIdx.x % kGroup;
  // the file this group slot belongs to: the last one whose first slot is <= slot
  int lo = 0, hi = n_files - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (files[mid].group_off <= slot) lo = mid; else hi = mid - 1;
  }
  const ResampleFile f = files[lo];
  const int64_t local = slot - f.group_off;
  if (local < 0 || local >= n_groups[lo]) return;
  const ResampleGroup g = groups[slot];
  if (k >= g.count) return;
  const float* in = reinterpret_cast<const float*>(raw + f.mono_off);
  float* out = reinterpret_cast<float*>(raw + f.out_off);
  const int64_t n = f.n_in;
  const double factor = f.factor;
  const double dt = 1.0 / factor;
  double t = g.t0;
  for (int j = 0; j < k; ++j) t += dt;                           // the converter's clock, step by step
  const double left_phase = t - floor(t), right_phase = 1.0 - left_phase;
  const int64_t c = (int64_t)g.base + (int64_t)(int)t;
  float v = 0.0f, w = 0.0f;
  if (factor >= 1) {                                             // lrsFilterUp, filterkit.c:115-170
    double ph = left_phase * kNpc;
    int h = (int)ph;
    for (int64_t i = 0; h < kNwing; h += kNpc, ++i) { const float p = imp[h] * rs_at(in, n, c - i); v = v + p; }
    ph = right_phase * kNpc;
    h = (int)ph;
    if (ph == 0) h += kNpc;
    for (int64_t i = 0; h < kNwing - 1; h += kNpc, ++i) { const float p = imp[h] * rs_at(in, n, c + 1 + i); w = w + p; }
  } else {                                                       // lrsFilterUD, filterkit.c:172-215
    const double dh = fmin((double)kNpc, factor * kNpc);         // resamplesubs.c:91
    double ho = left_phase * dh;
    for (int64_t i = 0; (int)ho < kNwing; ho += dh, ++i) { const float p = imp[(int)ho] * rs_at(in, n, c - i); v = v + p; }
    ho = right_phase * dh;
    if (right_phase == 0) ho += dh;
    for (int64_t i = 0; (int)ho < kNwing - 1; ho += dh, ++i) { const float p = imp[(int)ho] * rs_at(in, n, c + 1 + i); w = w + p; }
  }
  v = v + w;
  float lpscl = 1.0f;
  if (factor < 1) lpscl = (float)((double)lpscl * factor);       // resample.c:211-212
  out[g.out0 + k] = v * lpscl;                                   // resamplesubs.c:58
}

}  // namespace

hipError_t launch_resample(unsigned char* raw, const ResampleFile* files, int n_files, int64_t group_slots, int64_t max_n_in,
                           ResampleGroup* groups, int32_t* n_groups, const float* imp, hipStream_t stream) {
  if (n_files <= 0) return hipSuccess;
  const int per_file = (int)std::min<int64_t>(std::max<int64_t>((max_n_in + 256 * 8 - 1) / (256 * 8), 1), n_files >= 256 ? 4 : 256);
  resample_mix_kernel<<<dim3((unsigned)n_files, (unsigned)per_file), 256, 0, stream>>>(raw, files);
  resample_plan_kernel<<<(n_files + 63) / 64, 64, 0, stream>>>(raw, files, n_files, groups, n_groups);
  const int64_t blocks = (group_slots + (256 / kGroup) - 1) / (256 / kGroup);
  if (blocks > 0)
    resample_filter_kernel<<<(unsigned)blocks, 256, 0, stream>>>(raw, files, n_files, groups, n_groups, imp);
  return hipGetLastError();
}

}  // namespace afx
