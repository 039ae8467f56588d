// tools/ubench/hbm_stream.hip -- achievable HBM bandwidth on this MI355X with the access pattern of the frame kernels:
// a streaming read of f32 PCM (16 B per lane, grid-stride) with a tiny write per 4 KiB ("read"), and a plain copy.
// "measure achievable with a copy kernel and use that as the denominator too".
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/hbm_stream.hip -o /tmp/hbm_stream && /tmp/hbm_stream
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void read_kernel(const float4* __restrict__ src, size_t n, float* __restrict__ out) {
  float acc = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = src[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 12345.678f) out[0] = acc;   // never true: keeps the loads alive without a write per thread
}
__global__ __launch_bounds__(256) void copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

int main() {
  const size_t bytes = (size_t)4 << 30;   // 4 GiB: far beyond the 256 MiB of Infinity Cache
  const size_t n = bytes / 16;
  float4 *a = nullptr, *b = nullptr;
  float* out = nullptr;
  if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
  hipMemset(a, 1, bytes);
  hipMemset(b, 0, bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int blocks : {256 * 4, 256 * 8, 256 * 16, 256 * 32}) {
    for (int mode = 0; mode < 2; ++mode) {
      float best = 1e30f;
      for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(read_kernel, dim3(blocks), dim3(256), 0, 0, a, n, out);
        else hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, 0, a, b, n);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
      }
      const double moved = (mode == 0 ? 1.0 : 2.0) * (double)bytes;
      std::printf("%s, %5d workgroups of 256: %.3f ms, %.0f GB/s %s\n", mode == 0 ? "read 4 GiB" : "copy 4 GiB", blocks, best,
                  moved / (best * 1e-3) / 1e9, mode == 0 ? "(read)" : "(read + write)");
    }
  }
  return 0;
}
