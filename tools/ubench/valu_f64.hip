// Micro-benchmark: f64 / f32 VALU issue rate per SIMD as a function of waves per SIMD and ILP.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_f64 valu_f64.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <typename T, int ILP>
__global__ void fma_loop(T* out, int iters, T a, T b) {
  T x[ILP];
#pragma unroll
  for (int i = 0; i < ILP; ++i) x[i] = (T)threadIdx.x + (T)i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k)
#pragma unroll
      for (int i = 0; i < ILP; ++i) x[i] = x[i] * a + b;
  }
  T s = 0;
#pragma unroll
  for (int i = 0; i < ILP; ++i) s += x[i];
  if (s == (T)12345.678) out[0] = s;
}
template <typename T, int ILP>
__global__ void add_loop(T* out, int iters, T a, T b) {
  T x[ILP];
#pragma unroll
  for (int i = 0; i < ILP; ++i) x[i] = (T)threadIdx.x + (T)i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k)
#pragma unroll
      for (int i = 0; i < ILP; ++i) x[i] = x[i] + a;
  }
  T s = 0;
#pragma unroll
  for (int i = 0; i < ILP; ++i) s += x[i];
  if (s == (T)12345.678) out[0] = s;
}

template <typename K>
double time_kernel(K launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0);
  launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  double* d; hipMalloc(&d, 64);
  const int iters = 20000;
  const int cus = 256;
  printf("waves/SIMD  kernel        ms     cycles/instr/SIMD (at 2.1 GHz; 256 CUs x 4 SIMDs)\n");
  for (int wps = 1; wps <= 4; ++wps) {
    const int threads = 256 * wps;   // wps waves on each of the 4 SIMDs of a CU, 1 block per CU
    auto report = [&](const char* name, double ms, int ilp) {
      const double instr_per_wave = (double)iters * 16 * ilp;
      const double cyc = ms * 1e-3 * 2.1e9;
      printf("%d           %-12s %7.3f  %6.2f\n", wps, name, ms, cyc / (instr_per_wave * wps));
    };
    report("fma_f64 ilp8", time_kernel([&] { hipLaunchKernelGGL((fma_loop<double, 8>), dim3(cus), dim3(threads), 0, 0, d, iters, 1.0000001, 1e-9); }), 8);
    report("add_f64 ilp8", time_kernel([&] { hipLaunchKernelGGL((add_loop<double, 8>), dim3(cus), dim3(threads), 0, 0, d, iters, 1.0000001, 1e-9); }), 8);
    report("fma_f64 ilp2", time_kernel([&] { hipLaunchKernelGGL((fma_loop<double, 2>), dim3(cus), dim3(threads), 0, 0, d, iters, 1.0000001, 1e-9); }), 2);
    report("fma_f32 ilp8", time_kernel([&] { hipLaunchKernelGGL((fma_loop<float, 8>), dim3(cus), dim3(threads), 0, 0, (float*)d, iters, 1.0000001f, 1e-9f); }), 8);
    report("add_f32 ilp8", time_kernel([&] { hipLaunchKernelGGL((add_loop<float, 8>), dim3(cus), dim3(threads), 0, 0, (float*)d, iters, 1.0000001f, 1e-9f); }), 8);
  }
  return 0;
}
