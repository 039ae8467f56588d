// Micro-benchmark: does v_mfma_f64_4x4x4_4b co-execute with f64 VALU work?  Three loops per wave:
//   valu   : 64 independent v_fma_f64 per iteration
//   mfma   : 16 v_mfma_f64_4x4x4 per iteration (4 accumulator chains)
//   both   : the two interleaved
// Reported: wall cycles per iteration per SIMD at 1 and 2 waves per SIMD (256 CUs, all SIMDs busy).
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void loop(double* out, int iters) {
  double x[8], acc[4], a = threadIdx.x * 0.001 + 1.0, b = 0.5 - threadIdx.x * 0.002;
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x + i;
  for (int i = 0; i < 4; ++i) acc[i] = 0.0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (MODE & 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = x[i] * 1.0000001 + 1e-9;
      }
      if (MODE & 2) {
        acc[(2 * k) & 3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[(2 * k) & 3], 0, 0, 0);
        acc[(2 * k + 1) & 3] = __builtin_amdgcn_mfma_f64_4x4x4f64(b, a, acc[(2 * k + 1) & 3], 0, 0, 0);
      }
    }
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += x[i];
  for (int i = 0; i < 4; ++i) s += acc[i];
  if (s == 1.2345e-300) out[0] = s;
}

template <int MODE>
void run(double* d, int wps, const char* name) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(loop<MODE>, dim3(256), dim3(256 * wps), 0, 0, d, iters);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(loop<MODE>, dim3(256), dim3(256 * wps), 0, 0, d, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-5s waves/SIMD %d: %8.1f cycles per iteration per wave (64 fma + 16 mfma where present), %.3f ms\n", name, wps,
         ms * 1e-3 * 2.1e9 / iters, ms);
}

int main() {
  double* d; (void)hipMalloc(&d, 64);
  for (int wps = 1; wps <= 2; ++wps) {
    run<1>(d, wps, "valu");
    run<2>(d, wps, "mfma");
    run<3>(d, wps, "both");
  }
  return 0;
}
