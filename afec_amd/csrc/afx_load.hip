// afec_amd/csrc/afx_load.hip -- LoadSample front end on the GPU.
//
// TSampleAnalyser::LoadSample (SampleAnalyser.cpp:484-718) after the container decode: conversion to
// the "16-bit float" range, mono mix-down, rms / peak, peak normalisation factor, -48 dB silence trim
// and zero padding.  Byte / integer work and two reductions per file: HBM-bound by construction
// (2..4 bytes in, 8 bytes out per sample).
//
//   load_scan   max |x| and sum (x/32768)^2 of the mono mix, then first and last sample whose normalised
//               magnitude exceeds the silence floor (three small kernels, several workgroups per file)
//   load_write  writes scaling * x[lead + n] as doubles behind start_pad zeros (the arena is pre-zeroed)

#include <hip/hip_runtime.h>

#include "afx_internal.h"

namespace afx {
namespace {

// the reference's decoders hand LoadSample "16-bit floats" (CoreFileFormats/Export/SampleConverter.h)
__device__ __forceinline__ float to_16bit_float(const unsigned char* raw, int format, int64_t idx) {
  if (format == 0) return (float)reinterpret_cast<const short*>(raw)[idx];            // :446-449
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
// stage 3: first / last sample above the silence floor, merged with atomicMin / atomicMax.
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
  for (int64_t n = (int64_t)blockIdx.y * kLoadThreads + threadIdx.x; n < f.n_frames; n += (int64_t)gridDim.y * kLoadThreads) {
    const float x = mono_sample(src, f.format, f.channels, n);
    const double t = (double)(x / 32768.0f);
    sum_sq += t * t;
    mx = fmaxf(mx, fabsf(x));
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
  __shared__ long long sl[kLoadThreads / 64];
  const LoadFile f = files[blockIdx.x];
  const unsigned char* src = raw + f.raw_off;
  const double amplification = scan[blockIdx.x].amplification;
  // silent leading / trailing samples (SampleAnalyser.cpp:651-669)
  long long first = 0x7FFFFFFF, last = -1;
  for (int64_t n = (int64_t)blockIdx.y * kLoadThreads + threadIdx.x; n < f.n_frames; n += (int64_t)gridDim.y * kLoadThreads) {
    const float x = mono_sample(src, f.format, f.channels, n);
    if (fabs(amplification * (double)x) > silence_floor) {
      first = n < first ? n : first;
      last = n > last ? n : last;
    }
  }
  first = block_reduce(first, sl, [](long long a, long long b) { return a < b ? a : b; });
  last = block_reduce(last, sl, [](long long a, long long b) { return a > b ? a : b; });
  if (threadIdx.x == 0 && last >= 0) {
    atomicMin(&scan[blockIdx.x].lead, (int32_t)first);
    atomicMax(&scan[blockIdx.x].trail, (int32_t)last);
  }
}

__global__ __launch_bounds__(kLoadThreads) void load_write_kernel(const unsigned char* raw, const LoadFile* files,
                                                                  const LoadPlace* place, double* arena) {
  const LoadFile f = files[blockIdx.x];
  const LoadPlace p = place[blockIdx.x];
  if (f.n_frames <= 0 || p.out_n <= 0) return;
  const unsigned char* src = raw + f.raw_off;
  double* dst = arena + p.out_off;
  // only the analysed prefix is kept; zeros of the start / end pads are already there
  const int64_t n_copy = (p.audible < p.out_n - p.start_pad) ? p.audible : (p.out_n - p.start_pad);
  for (int64_t n = (int64_t)blockIdx.y * kLoadThreads + threadIdx.x; n < n_copy; n += (int64_t)gridDim.y * kLoadThreads)
    dst[p.start_pad + n] = (double)mono_sample(src, f.format, f.channels, p.lead + n) * p.scaling;  // SA:712-718
}

// CalcEffectiveLength (SampleAnalyser.cpp:1715-1755) on the normalised buffer: for each of three floors the
// first and the last sample above it.  out[buffer][6] = (first, last) x 3, initialised to (INT_MAX, -1);
// a buffer is scanned by gridDim.y workgroups that merge with atomicMin / atomicMax.
__global__ void effective_length_init_kernel(int32_t* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (i & 1) ? -1 : 0x7FFFFFFF;
}

template <typename TIn>
__global__ __launch_bounds__(kLoadThreads) void effective_length_kernel(const TIn* pcm, const BufSpan* spans, double f0,
                                                                        double f1, double f2, int32_t* out) {
  __shared__ long long sl[kLoadThreads / 64];
  const BufSpan sp = spans[blockIdx.x];
  const TIn* const x = pcm + sp.off;
  const double floors[3] = {f0, f1, f2};
  long long first[3] = {0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF}, last[3] = {-1, -1, -1};
  for (int64_t n = (int64_t)blockIdx.y * kLoadThreads + threadIdx.x; n < sp.n; n += (int64_t)gridDim.y * kLoadThreads) {
    const double v = fabs((double)x[n]);
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

}  // namespace

hipError_t launch_effective_length(const void* pcm, int pcm_dtype, const BufSpan* spans, int n_bufs, double floor48,
                                   double floor24, double floor12, int32_t* out, hipStream_t stream) {
  if (n_bufs <= 0) return hipSuccess;
  hipLaunchKernelGGL(effective_length_init_kernel, dim3((6 * n_bufs + 255) / 256), dim3(256), 0, stream, out, 6 * n_bufs);
  // few buffers: many workgroups per buffer; many buffers: one each
  const int per_buffer = n_bufs >= 1024 ? 1 : (n_bufs >= 64 ? 8 : 128);
  if (pcm_dtype == 0)
    hipLaunchKernelGGL(effective_length_kernel<float>, dim3(n_bufs, per_buffer), dim3(kLoadThreads), 0, stream,
                       reinterpret_cast<const float*>(pcm), spans, floor48, floor24, floor12, out);
  else
    hipLaunchKernelGGL(effective_length_kernel<double>, dim3(n_bufs, per_buffer), dim3(kLoadThreads), 0, stream,
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
  hipLaunchKernelGGL(load_scan3_kernel, dim3(n_files, per_file), dim3(kLoadThreads), 0, stream, raw, files, silence_floor, scan);
  return hipGetLastError();
}

hipError_t launch_load_write(const unsigned char* raw, const LoadFile* files, const LoadPlace* place, int n_files,
                             double* arena, hipStream_t stream) {
  if (n_files <= 0) return hipSuccess;
  hipLaunchKernelGGL(load_write_kernel, dim3(n_files, 8), dim3(kLoadThreads), 0, stream, raw, files, place, arena);
  return hipGetLastError();
}

}  // namespace afx
