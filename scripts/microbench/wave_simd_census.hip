// Which SIMD does wave w of a 1024-thread workgroup run on?  (gfx950: HW_REG_HW_ID bits 5:4 = SIMD_ID, 3:0 = WAVE_ID, 11:8 = CU_ID)
//   hipcc --offload-arch=gfx950 -O3 wave_simd_census.hip -o /tmp/census && /tmp/census
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ __launch_bounds__(1024) void census(unsigned* out) {
  extern __shared__ float lds[];
  lds[threadIdx.x] = 1.f;
  __syncthreads();
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = id;
}
int main() {
  unsigned* d;
  CK(hipMalloc(&d, 256 * 16 * 4));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(census), hipFuncAttributeMaxDynamicSharedMemorySize, 147 * 1024));
  hipLaunchKernelGGL(census, dim3(256), dim3(1024), 147 * 1024, 0, d);
  CK(hipDeviceSynchronize());
  unsigned h[256 * 16];
  CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
  int hist[16][4] = {};
  for (int b = 0; b < 256; ++b)
    for (int w = 0; w < 16; ++w) hist[w][(h[b * 16 + w] >> 4) & 3]++;
  for (int w = 0; w < 16; ++w) printf("wave %2d: SIMD0 %3d  SIMD1 %3d  SIMD2 %3d  SIMD3 %3d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
  for (int b = 0; b < 4; ++b) {
    printf("block %d simd of waves 0..15:", b);
    for (int w = 0; w < 16; ++w) printf(" %u", (h[b * 16 + w] >> 4) & 3);
    printf("\n");
  }
  return 0;
}
