// which way do the gfx9 wavefront-wide DPP shifts move data on gfx950?  prints, for lanes 0, 1, 62, 63, the lane whose value each lane received
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  const int v = threadIdx.x;
  out[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
  out[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x134 /* wave_rol:1 */, 0xF, 0xF, false);
  out[128 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
  out[192 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x13C /* wave_ror:1 */, 0xF, 0xF, false);
}
int main() {
  int* d; int h[256];
  hipMalloc(&d, sizeof h);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const char* names[4] = {"wave_shl:1", "wave_rol:1", "wave_shr:1", "wave_ror:1"};
  for (int i = 0; i < 4; ++i)
    printf("%s: lane0<-%d lane1<-%d lane15<-%d lane16<-%d lane31<-%d lane32<-%d lane62<-%d lane63<-%d\n", names[i], h[64*i], h[64*i+1], h[64*i+15], h[64*i+16], h[64*i+31], h[64*i+32], h[64*i+62], h[64*i+63]);
  return 0;
}
