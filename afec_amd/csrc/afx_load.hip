// afec_amd/csrc/afx_load.hip -- LoadSample front end on the GPU.
//
// TSampleAnalyser::LoadSample (SampleAnalyser.cpp:484-718) after the container decode: conversion to
// the "16-bit float" range, mono mix-down, rms / peak, peak normalisation factor, -48 dB silence trim
// and zero padding.  Byte / integer work and two reductions per file: HBM-bound by construction
// (2..4 bytes in, 4 bytes out per sample).
//
//   load_scan   max |x| and sum (x/32768)^2 of the mono mix (several workgroups per file), then first and last sample
//               whose normalised magnitude exceeds the silence floor (one workgroup per file, from both ends)
//   load_write  writes the float mono signal x[lead + n] behind start_pad zeros, and the zeros of the pads; the
//               buffer's FinalScaling stays a per-buffer double (kPcmScaledF32, afx_internal.h): the reference's
//               mData[n] = (double)x * FinalScaling (SA:710-718) is formed by every consumer as it loads a sample

#include <hip/hip_runtime.h>

#include "afx_internal.h"
#include "afx_device.h"

namespace afx {
namespace {

// the reference's decoders hand LoadSample "16-bit floats" (CoreFileFormats/Export/SampleConverter.h)
__device__ __forceinline__ float to_16bit_float(const unsigned char* raw, int format, int64_t idx) {
  if (format == 0) return (float)reinterpret_cast<const short*>(raw)[idx];            // :446-449
  if (format == kRawMonoFloat) return reinterpret_cast<const float*>(raw)[idx];       // converted samples (afx_resample.hip)
  if (format == 1) {                                                                  // :474-486
    const unsigned char* b = raw + 3 * idx;
    const int v = (int)(((unsigned)b[0] | ((unsigned)b[1] << 8) | ((unsigned)b[2] << 16)) << 8);
    return (float)((double)v * 32768.0 / 2147483648.0);
  }
  if (format == 3) {                                                                  // :514-518 (32-bit signed)
    const float v = (float)((double)reinterpret_cast<const int*>(raw)[idx] * 32768.0 / 2147483648.0);
    return fmaxf(-32768.0f, fminf(32767.0f, v));
  }
  // normalised floats, 32 or 64 bit (S0To1FloatTo16BitFloat takes a double)                 :529-533
  const double d = (format == 4 ? reinterpret_cast<const double*>(raw)[idx] : (double)reinterpret_cast<const float*>(raw)[idx]) * 32768.0;
  return (float)(d < -32768.0 ? -32768.0 : (d > 32767.0 ? 32767.0 : d));
}

// mono mix in float with the first channel as destination (SampleAnalyser.cpp:535-548)
__device__ __forceinline__ float mono_sample(const unsigned char* raw, int format, int channels, int64_t n) {
  float d = to_16bit_float(raw, format, n * channels);
  if (channels > 1) {
    for (int c = 1; c < channels; ++c) d = __fadd_rn(d, to_16bit_float(raw, format, n * channels + c));
    d = __fmul_rn(d, 1.0f / (float)channels);
  }
  return d;
}

// the mono mix of frames n4 .. n4 + 3 (n4 a multiple of 4; frames at or behind n_frames come back as 0).  16-bit files
// -- the common case -- take one 8- or 16-byte load (a file's payload starts on a 16-byte boundary of the staging
// arena, afx_batch_create.cpp); every other format and channel count goes sample by sample.  The arithmetic is mono_sample's.
__device__ __forceinline__ void mono4(const unsigned char* raw, int format, int channels, int64_t n4, int64_t n_frames, float (&out)[4]) {
  if (format == 0 && channels <= 2 && n4 + 4 <= n_frames) {
    if (channels == 1) {
      const uint2 w = *reinterpret_cast<const uint2*>(raw + 2 * n4);
      out[0] = (float)(short)(w.x & 0xFFFFu); out[1] = (float)(short)(w.x >> 16);
      out[2] = (float)(short)(w.y & 0xFFFFu); out[3] = (float)(short)(w.y >> 16);
    } else {
      const uint4 w = *reinterpret_cast<const uint4*>(raw + 4 * n4);
      const unsigned v[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
      for (int k = 0; k < 4; ++k)
        out[k] = __fmul_rn(__fadd_rn((float)(short)(v[k] & 0xFFFFu), (float)(short)(v[k] >> 16)), 1.0f / 2.0f);
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) out[k] = (n4 + k < n_frames) ? mono_sample(raw, format, channels, n4 + k) : 0.0f;
}

constexpr int kLoadThreads = 256;

template <typename T, typename Op>
__device__ __forceinline__ T block_reduce(T v, T* scratch, Op op) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = op(v, __shfl_xor(v, o));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
  __syncthreads();
  T r = scratch[0];
  for (int w = 1; w < kLoadThreads / 64; ++w) r = op(r, scratch[w]);
  return r;
}

// A file is scanned by gridDim.y workgroups (many when the batch has few files, one when it has thousands).
// Stage 1: per-workgroup partial sum of squares and maximum; stage 2 (one workgroup per file) adds the
// partials in index order -- the result does not depend on timing -- and derives the amplification;
// stage 3: first / last sample above the silence floor (load_scan3_kernel).
struct LoadPartial {
  double sum_sq;
  float max_amp;
  float pad;
};

__global__ __launch_bounds__(kLoadThreads) void load_scan1_kernel(const unsigned char* raw, const LoadFile* files,
                                                                  LoadPartial* partial) {
  __shared__ double sd[kLoadThreads / 64];
  __shared__ float sf[kLoadThreads / 64];
  const LoadFile f = files[blockIdx.x];
  const unsigned char* src = raw + f.raw_off;
  // rms and peak (SampleAnalyser.cpp:612-637)
  double sum_sq = 0.0;
  float mx = 0.0f;
  for (int64_t n = 4 * ((int64_t)blockIdx.y * kLoadThreads + threadIdx.x); n < f.n_frames; n += 4 * (int64_t)gridDim.y * kLoadThreads) {
    float x[4];
    mono4(src, f.format, f.channels, n, f.n_frames, x);   // frames behind the end are 0: they change neither result
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const double t = (double)(x[k] / 32768.0f);
      sum_sq += t * t;
      mx = fmaxf(mx, fabsf(x[k]));
    }
  }
  sum_sq = block_reduce(sum_sq, sd, [](double a, double b) { return a + b; });
  mx = block_reduce(mx, sf, [](float a, float b) { return fmaxf(a, b); });
  if (threadIdx.x == 0) partial[(int64_t)blockIdx.x * gridDim.y + blockIdx.y] = LoadPartial{sum_sq, mx, 0.0f};
}

__global__ void load_scan2_kernel(const LoadFile* files, const LoadPartial* partial, int per_file, int n_files, LoadScan* scan) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_files) return;
  double sum_sq = 0.0;
  float mx = 0.0f;
  if (files[i].n_frames > 0)
    for (int y = 0; y < per_file; ++y) {
      sum_sq += partial[(int64_t)i * per_file + y].sum_sq;
      mx = fmaxf(mx, partial[(int64_t)i * per_file + y].max_amp);
    }
  LoadScan r;
  r.sum_sq = sum_sq;
  r.amplification = ((double)mx > (double)1e-12f) ? 32768.0 / (double)mx : 1.0;   // SampleAnalyser.cpp:639-640
  r.max_amp = mx;
  r.lead = 0x7FFFFFFF;      // first sample above the floor (stage 3)
  r.trail = -1;             // last sample above the floor
  r.pad = 0;
  scan[i] = r;
}

__global__ __launch_bounds__(kLoadThreads) void load_scan3_kernel(const unsigned char* raw, const LoadFile* files,
                                                                  double silence_floor, LoadScan* scan) {
  // silent leading / trailing samples (SampleAnalyser.cpp:651-669): one workgroup per file walks blocks of 1024 frames
  // from the front until one holds a sample above the floor, then from the back -- like the reference's two loops it
  // reads the silence and little else (a file without silence: 8 KiB of it)
  __shared__ long long sl[kLoadThreads / 64];
  const LoadFile f = files[blockIdx.x];
  if (f.n_frames <= 0) return;
  const unsigned char* src = raw + f.raw_off;
  const double amplification = scan[blockIdx.x].amplification;
  constexpr int64_t kBlock = 4 * kLoadThreads;
  const int64_t n_blocks = (f.n_frames + kBlock - 1) / kBlock;
  auto block_hits = [&](int64_t b, long long& first, long long& last) {
    float x[4];
    const int64_t n4 = b * kBlock + 4 * (int64_t)threadIdx.x;
    first = 0x7FFFFFFF; last = -1;
    if (n4 < f.n_frames) {
      mono4(src, f.format, f.channels, n4, f.n_frames, x);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (n4 + k < f.n_frames && fabs(amplification * (double)x[k]) > silence_floor) {
          first = first < n4 + k ? first : n4 + k;
          last = n4 + k;
        }
    }
    return __syncthreads_or(last >= 0) != 0;
  };
  long long first = 0x7FFFFFFF, last = -1, lo, hi;
  int64_t b = 0;
  for (; b < n_blocks; ++b)
    if (block_hits(b, lo, hi)) {
      first = block_reduce(lo, sl, [](long long p, long long q) { return p < q ? p : q; });
      break;
    }
  if (b == n_blocks) return;                 // nothing above the floor: lead / trail keep their initial values
  for (int64_t e = n_blocks - 1; e >= b; --e)
    if (block_hits(e, lo, hi)) {
      last = block_reduce(hi, sl, [](long long p, long long q) { return p > q ? p : q; });
      break;
    }
  if (threadIdx.x == 0) {
    scan[blockIdx.x].lead = (int32_t)first;
    scan[blockIdx.x].trail = (int32_t)last;
  }
}

__global__ __launch_bounds__(kLoadThreads) void load_write_kernel(const unsigned char* raw, const LoadFile* files,
                                                                  const LoadPlace* place, float* arena) {
  const LoadFile f = files[blockIdx.x];
  const LoadPlace p = place[blockIdx.x];
  if (f.n_frames <= 0 || p.out_n <= 0) return;
  const unsigned char* src = raw + f.raw_off;
  float* dst = arena + p.out_off;
  // only the analysed prefix is kept
  int64_t n_copy = (p.audible < p.out_n - p.start_pad) ? p.audible : (p.out_n - p.start_pad);
  if (n_copy < 0) n_copy = 0;
  // the start pad, the end pad and the slack up to the buffer's 4-sample slot (afx_batch_plan.cpp, place_buffers) are zeros:
  // written here, by the first workgroup of the file, so that the arena needs no memset (it is written exactly once)
  if (blockIdx.y == 0) {
    const int64_t slot = (p.out_n + 3) & ~(int64_t)3, tail = p.start_pad + n_copy;
    const int64_t pad = p.start_pad < slot ? p.start_pad : slot;   // a capped analysis may end inside the start pad
    for (int64_t n = threadIdx.x; n < pad; n += kLoadThreads) dst[n] = 0.0f;
    for (int64_t n = tail + threadIdx.x; n < slot; n += kLoadThreads) dst[n] = 0.0f;
  }
  // groups of four source frames on the source's alignment (vector loads); frames before `lead` and behind the copied
  // range are skipped
  const int64_t s_end = p.lead + n_copy;
  for (int64_t s4 = (p.lead & ~(int64_t)3) + 4 * ((int64_t)blockIdx.y * kLoadThreads + threadIdx.x); s4 < s_end;
       s4 += 4 * (int64_t)gridDim.y * kLoadThreads) {
    float x[4];
    mono4(src, f.format, f.channels, s4, f.n_frames, x);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (s4 + k >= p.lead && s4 + k < s_end) dst[p.start_pad + (s4 + k - p.lead)] = x[k];  // x p.scaling on load: SA:712-718
  }
}

// CalcEffectiveLength (SampleAnalyser.cpp:1715-1755) on the normalised buffer: for each of three floors the
// first and the last sample above it.  out[buffer][6] = (first, last) x 3, initialised to (INT_MAX, -1);
// a buffer is scanned by gridDim.y workgroups that merge with atomicMin / atomicMax.
__global__ void effective_length_init_kernel(int32_t* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (i & 1) ? -1 : 0x7FFFFFFF;
}

template <typename TIn, bool SCALED>
__global__ __launch_bounds__(kLoadThreads) void effective_length_kernel(const TIn* pcm, const BufSpan* spans, double f0,
                                                                        double f1, double f2, int32_t* out) {
  __shared__ long long sl[kLoadThreads / 64];
  const BufSpan sp = spans[blockIdx.x];
  const TIn* const x = pcm + sp.off;
  const double floors[3] = {f0, f1, f2};
  long long first[3] = {0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF}, last[3] = {-1, -1, -1};
  for (int64_t n = (int64_t)blockIdx.y * kLoadThreads + threadIdx.x; n < sp.n; n += (int64_t)gridDim.y * kLoadThreads) {
    const double v = fabs(pcm_double<SCALED>(x[n], sp.scale));
#pragma unroll
    for (int j = 0; j < 3; ++j)
      if (v > floors[j]) {
        first[j] = n < first[j] ? n : first[j];
        last[j] = n > last[j] ? n : last[j];
      }
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const long long a = block_reduce(first[j], sl, [](long long p, long long q) { return p < q ? p : q; });
    const long long b = block_reduce(last[j], sl, [](long long p, long long q) { return p > q ? p : q; });
    if (threadIdx.x == 0) {
      if (a != 0x7FFFFFFF) atomicMin(&out[6 * blockIdx.x + 2 * j], (int32_t)a);
      if (b >= 0) atomicMax(&out[6 * blockIdx.x + 2 * j + 1], (int32_t)b);
    }
  }
}

// The same for batches of many buffers: one workgroup per buffer walks blocks of 1024 samples from the front until
// the highest floor has been crossed, then from the back -- typical audio is read only at its ends.
template <typename TIn, bool SCALED>
__global__ __launch_bounds__(kLoadThreads) void effective_length_ends_kernel(const TIn* pcm, const BufSpan* spans, double f0,
                                                                             double f1, double f2, int32_t* out) {
  __shared__ long long sl[kLoadThreads / 64];
  const BufSpan sp = spans[blockIdx.x];
  const TIn* const x = pcm + sp.off;
  const double floors[3] = {f0, f1, f2};   // ascending (-48, -24, -12 dB)
  constexpr int64_t kBlock = 4 * kLoadThreads;
  const int64_t n_blocks = (sp.n + kBlock - 1) / kBlock;
  long long first[3] = {0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF}, last[3] = {-1, -1, -1};
  // forward: a block's hits per floor; stop behind the block in which the highest floor was crossed
  for (int64_t b = 0; b < n_blocks; ++b) {
    long long lo[3] = {0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t n = b * kBlock + (int64_t)k * kLoadThreads + threadIdx.x;
      const double v = n < sp.n ? fabs(pcm_double<SCALED>(x[n], sp.scale)) : 0.0;
#pragma unroll
      for (int j = 0; j < 3; ++j)
        if (v > floors[j] && n < lo[j]) lo[j] = n;
    }
    if (!__syncthreads_or(lo[0] != 0x7FFFFFFF)) continue;   // nothing above the lowest floor in this block
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const long long a = block_reduce(lo[j], sl, [](long long p, long long q) { return p < q ? p : q; });
      if (first[j] == 0x7FFFFFFF) first[j] = a;
    }
    if (first[2] != 0x7FFFFFFF) break;
  }
  if (first[0] != 0x7FFFFFFF)
    for (int64_t b = n_blocks - 1; b >= 0; --b) {
      long long hi[3] = {-1, -1, -1};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int64_t n = b * kBlock + (int64_t)k * kLoadThreads + threadIdx.x;
        const double v = n < sp.n ? fabs(pcm_double<SCALED>(x[n], sp.scale)) : 0.0;
#pragma unroll
        for (int j = 0; j < 3; ++j)
          if (v > floors[j] && n > hi[j]) hi[j] = n;
      }
      if (!__syncthreads_or(hi[0] >= 0)) continue;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const long long a = block_reduce(hi[j], sl, [](long long p, long long q) { return p > q ? p : q; });
        if (last[j] < 0) last[j] = a;
      }
      // a floor that was never crossed from the front is not crossed from the back either
      if ((last[2] >= 0 || first[2] == 0x7FFFFFFF) && (last[1] >= 0 || first[1] == 0x7FFFFFFF)) break;
    }
  if (threadIdx.x == 0)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      out[6 * blockIdx.x + 2 * j] = (int32_t)first[j];
      out[6 * blockIdx.x + 2 * j + 1] = (int32_t)last[j];
    }
}

}  // namespace

hipError_t launch_effective_length(const void* pcm, int pcm_dtype, const BufSpan* spans, int n_bufs, double floor48,
                                   double floor24, double floor12, int32_t* out, hipStream_t stream) {
  if (n_bufs <= 0) return hipSuccess;
  if (n_bufs >= 64) {   // many buffers: one workgroup each, from both ends (writes every output itself)
    if (pcm_dtype == kPcmF32)
      hipLaunchKernelGGL((effective_length_ends_kernel<float, false>), dim3(n_bufs), dim3(kLoadThreads), 0, stream,
                         reinterpret_cast<const float*>(pcm), spans, floor48, floor24, floor12, out);
    else if (pcm_dtype == kPcmScaledF32)
      hipLaunchKernelGGL((effective_length_ends_kernel<float, true>), dim3(n_bufs), dim3(kLoadThreads), 0, stream,
                         reinterpret_cast<const float*>(pcm), spans, floor48, floor24, floor12, out);
    else
      hipLaunchKernelGGL((effective_length_ends_kernel<double, false>), dim3(n_bufs), dim3(kLoadThreads), 0, stream,
                         reinterpret_cast<const double*>(pcm), spans, floor48, floor24, floor12, out);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(effective_length_init_kernel, dim3((6 * n_bufs + 255) / 256), dim3(256), 0, stream, out, 6 * n_bufs);
  // few buffers: many workgroups per buffer; many buffers: one each
  const int per_buffer = 128;
  if (pcm_dtype == kPcmF32)
    hipLaunchKernelGGL((effective_length_kernel<float, false>), dim3(n_bufs, per_buffer), dim3(kLoadThreads), 0, stream,
                       reinterpret_cast<const float*>(pcm), spans, floor48, floor24, floor12, out);
  else if (pcm_dtype == kPcmScaledF32)
    hipLaunchKernelGGL((effective_length_kernel<float, true>), dim3(n_bufs, per_buffer), dim3(kLoadThreads), 0, stream,
                       reinterpret_cast<const float*>(pcm), spans, floor48, floor24, floor12, out);
  else
    hipLaunchKernelGGL((effective_length_kernel<double, false>), dim3(n_bufs, per_buffer), dim3(kLoadThreads), 0, stream,
                       reinterpret_cast<const double*>(pcm), spans, floor48, floor24, floor12, out);
  return hipGetLastError();
}

int load_scan_blocks_per_file(int n_files) { return n_files >= 1024 ? 1 : (n_files >= 64 ? 8 : 64); }

hipError_t launch_load_scan(const unsigned char* raw, const LoadFile* files, int n_files, double silence_floor,
                            void* partial_scratch, LoadScan* scan, hipStream_t stream) {
  if (n_files <= 0) return hipSuccess;
  const int per_file = load_scan_blocks_per_file(n_files);
  LoadPartial* partial = reinterpret_cast<LoadPartial*>(partial_scratch);
  hipLaunchKernelGGL(load_scan1_kernel, dim3(n_files, per_file), dim3(kLoadThreads), 0, stream, raw, files, partial);
  hipLaunchKernelGGL(load_scan2_kernel, dim3((n_files + 255) / 256), dim3(256), 0, stream, files, partial, per_file, n_files, scan);
  hipLaunchKernelGGL(load_scan3_kernel, dim3(n_files), dim3(kLoadThreads), 0, stream, raw, files, silence_floor, scan);
  return hipGetLastError();
}

hipError_t launch_load_write(const unsigned char* raw, const LoadFile* files, const LoadPlace* place, int n_files,
                             float* arena, hipStream_t stream) {
  if (n_files <= 0) return hipSuccess;
  hipLaunchKernelGGL(load_write_kernel, dim3(n_files, 8), dim3(kLoadThreads), 0, stream, raw, files, place, arena);
  return hipGetLastError();
}

}  // namespace afx
