// Microbenchmark: fp32 MFMA alone, fp32 VALU FMA alone, both in one wave, both in separate waves (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>  // 0 mfma, 1 valu, 2 both same wave, 3 split by wave parity
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  f32x2 v[32];
  for (int i = 0; i < 32; ++i) v[i] = f32x2{(float)i, (float)lane};
  float a = a0 + lane * 1e-6f, b = b0;
  f32x2 pa = {a, a * 0.5f}, pb = {b, b * 0.25f};
  const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && (wave & 1) == 0);
  const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && (wave & 1) == 1);
  for (int it = 0; it < iters; ++it) {
    if (do_m) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    if (do_v) {
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int i = 0; i < 32; ++i) v[i] = __builtin_elementwise_fma(pa, pb, v[i]);   // v_pk_fma_f32: 2 FMA / lane
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 32; ++i) s += v[i][0] + v[i][1];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE> void run(const char* name, float* d, int blocks) {
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 100, 1.0f, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double waves = (double)blocks * 4;
  double mf = 0, vf = 0;
  if (MODE == 0 || MODE == 2) mf = waves * iters * 16.0 * 4096.0;            // 16 MFMA x 2*32*32*2 flop
  if (MODE == 3) mf = waves / 2 * iters * 16.0 * 4096.0;
  if (MODE == 1 || MODE == 2) vf = waves * iters * 256.0 * 64 * 4.0;         // 256 pk_fma x 64 lanes x 2 FMA x 2 flop
  if (MODE == 3) vf = waves / 2 * iters * 256.0 * 64 * 4.0;
  printf("%-28s blocks/CU=%d  %8.3f ms  mfma %7.1f TF  valu %7.1f TF  total %7.1f TF\n", name, blocks / 256, ms, mf / ms / 1e9,
         vf / ms / 1e9, (mf + vf) / ms / 1e9);
}

int main() {
  float* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
  for (int bpc = 1; bpc <= 2; ++bpc) {
    const int blocks = 256 * bpc;
    run<0>("mfma only", d, blocks);
    run<1>("valu (v_pk_fma) only", d, blocks);
    run<2>("both, same wave", d, blocks);
    run<3>("both, alternate waves", d, blocks);
  }
  return 0;
}
