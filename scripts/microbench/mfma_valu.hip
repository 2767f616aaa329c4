// Microbenchmark (gfx950): what does a vector instruction cost a co-resident fp32 MFMA stream?
// fp32 MFMA alone, a VALU stream alone, both in one wave, both in alternate waves - for four kinds of VALU work:
// packed fp32 FMA, plain fp32 FMA, the transcendental v_exp_f32, and 32-bit integer adds.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// VOP: 0 v_pk_fma_f32, 1 v_fma_f32, 2 v_exp_f32, 3 v_add_u32
template <int VOP>
__device__ __forceinline__ void valu_block(f32x2 (&v)[32], f32x2 pa, f32x2 pb) {
#pragma unroll
  for (int s = 0; s < 8; ++s)
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      if (VOP == 0) v[i] = __builtin_elementwise_fma(pa, pb, v[i]);
      if (VOP == 1) v[i].x = __builtin_fmaf(pa.x, pb.x, v[i].x);
      if (VOP == 2) v[i].x = __builtin_amdgcn_exp2f(v[i].x);
      if (VOP == 3) v[i].x = __int_as_float(__float_as_int(v[i].x) + __float_as_int(pa.x));
    }
}

template <int MODE, int VOP>  // MODE: 0 mfma, 1 valu, 2 both same wave, 3 split by wave parity
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  f32x2 v[32];
  for (int i = 0; i < 32; ++i) v[i] = f32x2{(float)i * 1e-3f, (float)lane};
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
    if (do_v) valu_block<VOP>(v, pa, pb);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 32; ++i) s += v[i][0] + v[i][1];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int VOP> void run(const char* name, float* d, int blocks) {
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, VOP>), dim3(blocks), dim3(256), 0, 0, d, 100, 1.0f, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, VOP>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double waves = (double)blocks * 4;
  double mf = 0, vi = 0;   // MFMA TFLOP/s; VALU instructions per ns (wave-instructions over the chip)
  const double mwaves = (MODE == 3) ? waves / 2 : (MODE == 0 || MODE == 2 ? waves : 0);
  const double vwaves = (MODE == 3) ? waves / 2 : (MODE == 1 || MODE == 2 ? waves : 0);
  mf = mwaves * iters * 16.0 * 4096.0;
  vi = vwaves * iters * 256.0;
  printf("%-34s blocks/CU=%d  %8.3f ms  mfma %7.1f TF   valu %7.2f Ginstr/s (%5.2f cycles per wave-instr per SIMD)\n", name, blocks / 256, ms,
         mf / ms / 1e9, vi / ms / 1e6, vi > 0 ? (ms * 1e-3 * 2.4e9) / (vi / 1024.0) : 0.0);
}

template <int VOP> void suite(const char* vname, float* d) {
  char nm[64];
  const int blocks = 512;
  snprintf(nm, sizeof nm, "%s only", vname); run<1, VOP>(nm, d, blocks);
  snprintf(nm, sizeof nm, "mfma + %s, same wave", vname); run<2, VOP>(nm, d, blocks);
  snprintf(nm, sizeof nm, "mfma + %s, alternate waves", vname); run<3, VOP>(nm, d, blocks);
}

int main() {
  float* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
  run<0, 0>("mfma only", d, 512);
  suite<0>("v_pk_fma_f32", d);
  suite<1>("v_fma_f32", d);
  suite<2>("v_exp_f32", d);
  suite<3>("v_add_u32", d);
  return 0;
}
