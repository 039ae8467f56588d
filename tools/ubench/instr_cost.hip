// Micro-benchmark: wall cycles per wave-instruction for the cross-lane / LDS / conversion ops the
// frames kernel leans on, at 1..3 waves per SIMD (1 workgroup per CU).  Each loop body is a chain of
// 8 independent registers so that the issue rate, not the dependency latency, is measured.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(...) _Pragma("unroll") for (int rep_ = 0; rep_ < 16; ++rep_) { __VA_ARGS__ }

enum { K_SWAP32, K_SWAP16, K_DPP, K_BPERM, K_WR64, K_RD64, K_RD128, K_WR128, K_CVT, K_RSQ, K_CNDMASK, K_ADD64, K_COUNT };
static const char* kNames[] = {"permlane32_swap", "permlane16_swap", "mov_dpp", "ds_bpermute_b32", "ds_write_b64",
                               "ds_read_b64", "ds_read_b128", "ds_write_b128", "cvt_f64_f32", "rsq_f64", "cndmask_b32", "add_f64"};

template <int KIND>
__global__ void loop(double* out, int iters) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x;
  unsigned a[8];
  double d[8];
  for (int i = 0; i < 8; ++i) { a[i] = lane * 3 + i; d[i] = lane + i * 0.5; }
  double* p = lds + (threadIdx.x >> 6) * 2200 + (lane & 63);
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
    if constexpr (KIND == K_SWAP32) {
      REP16({ auto r = __builtin_amdgcn_permlane32_swap(a[0], a[1], false, false); a[0] = r[0]; a[1] = r[1];
              auto s = __builtin_amdgcn_permlane32_swap(a[2], a[3], false, false); a[2] = s[0]; a[3] = s[1]; })
    } else if constexpr (KIND == K_SWAP16) {
      REP16({ auto r = __builtin_amdgcn_permlane16_swap(a[0], a[1], false, false); a[0] = r[0]; a[1] = r[1];
              auto s = __builtin_amdgcn_permlane16_swap(a[2], a[3], false, false); a[2] = s[0]; a[3] = s[1]; })
    } else if constexpr (KIND == K_DPP) {
      REP16({ a[0] = __builtin_amdgcn_update_dpp(0, a[0], 0xB1, 0xF, 0xF, true); a[1] = __builtin_amdgcn_update_dpp(0, a[1], 0x4E, 0xF, 0xF, true); })
    } else if constexpr (KIND == K_BPERM) {
      REP16({ a[0] = __builtin_amdgcn_ds_bpermute((lane ^ 5) << 2, a[0]); a[1] = __builtin_amdgcn_ds_bpermute((lane ^ 9) << 2, a[1]); })
    } else if constexpr (KIND == K_WR64) {
      REP16({ p[0] = d[0]; p[65] = d[1]; __builtin_amdgcn_sched_barrier(0); })
    } else if constexpr (KIND == K_RD64) {
      REP16({ double x, y; asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:520" : "=v"(x), "=v"(y) : "v"((unsigned)(lane * 8))); asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x), "+v"(y)); d[0] += x; d[1] += y; })
    } else if constexpr (KIND == K_RD128) {
      using d2 = __attribute__((ext_vector_type(2))) double;
      REP16({ d2 x, y; asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1040" : "=v"(x), "=v"(y) : "v"((unsigned)(lane * 16))); asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x), "+v"(y)); d[0] += x.x; d[1] += y.y; })
    } else if constexpr (KIND == K_WR128) {
      using d2 = __attribute__((ext_vector_type(2))) double;
      d2* q = reinterpret_cast<d2*>(lds) + (threadIdx.x >> 6) * 1100 + (lane & 63);
      REP16({ d2 t; t.x = d[0]; t.y = d[1]; q[0] = t; q[66] = t; __builtin_amdgcn_sched_barrier(0); })
    } else if constexpr (KIND == K_CVT) {
      REP16({ d[0] = (double)__uint_as_float(a[0]) + d[0]; d[1] = (double)__uint_as_float(a[1]) + d[1]; })
    } else if constexpr (KIND == K_RSQ) {
      REP16({ d[0] = __builtin_amdgcn_rsq(d[0]); d[1] = __builtin_amdgcn_rsq(d[1]); })
    } else if constexpr (KIND == K_CNDMASK) {
      REP16({ a[0] = (lane & 1) ? a[1] : a[0]; a[2] = (lane & 2) ? a[3] : a[2]; asm volatile("" : "+v"(a[0]), "+v"(a[2])); })
    } else if constexpr (KIND == K_ADD64) {
      REP16({ d[0] += 1.5; d[1] += 2.5; d[2] += 0.5; d[3] += 0.25; })
    }
    acc += a[0] ^ a[1];
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += d[i] + a[i];
  if (s + acc == 1.234567e-300) out[0] = s;
}

template <int KIND>
void run(double* d, int wps) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const size_t lds = 4 * wps * 2200 * 8;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(loop<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(loop<KIND>, dim3(256), dim3(256 * wps), lds, 0, d, iters);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(loop<KIND>, dim3(256), dim3(256 * wps), lds, 0, d, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const int per_iter = (KIND == K_ADD64) ? 64 : 32;   // instructions of the measured kind per loop iteration
  const double cyc = ms * 1e-3 * 2.1e9 / ((double)iters * per_iter);
  printf("%-18s waves/SIMD %d: %7.2f cycles per wave-instruction (wall, all %d waves of the SIMD together: %6.2f per SIMD)\n",
         kNames[KIND], wps, cyc, wps, cyc / wps);
}

template <int KIND>
void sweep(double* d) {
  for (int wps = 1; wps <= 3; ++wps) run<KIND>(d, wps);
}

int main() {
  double* d; (void)hipMalloc(&d, 64);
  sweep<K_SWAP32>(d); sweep<K_SWAP16>(d); sweep<K_DPP>(d); sweep<K_BPERM>(d); sweep<K_WR64>(d); sweep<K_RD64>(d);
  sweep<K_RD128>(d); sweep<K_WR128>(d); sweep<K_CVT>(d); sweep<K_RSQ>(d); sweep<K_CNDMASK>(d); sweep<K_ADD64>(d);
  return 0;
}
