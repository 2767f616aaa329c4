// Does v_mfma_f32_32x32x16_f16 honour subnormal f16 inputs on gfx950?  A = 2^-20 (subnormal), B = 2^10: every output element
// is 16 * 2^-10 = 2^-6 if subnormals are read as they are, 0 if they are flushed.   hipcc --offload-arch=gfx950 -O2 mfma_denorm.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float a, float b) {
  f16x8 A, B;
  for (int e = 0; e < 8; ++e) { A[e] = (_Float16)a; B[e] = (_Float16)b; }
  f32x16 C;
  for (int r = 0; r < 16; ++r) C[r] = 0.f;
  C = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = C[0]; out[1] = (float)A[0]; }
}
int main() {
  float* d; hipMalloc(&d, 8);
  const float as[3] = {9.5367431640625e-07f /* 2^-20 */, 5.9604644775390625e-08f /* 2^-24, the smallest */, 6.103515625e-05f /* 2^-14, normal */};
  for (float a : as) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, a, 1024.f);
    float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("a = %.3e (as f16 %.3e) x 1024 x 16 -> %.6e (exact %.6e)\n", a, h[1], h[0], a * 1024.f * 16.f);
  }
  return 0;
}
