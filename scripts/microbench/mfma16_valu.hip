// Microbenchmark (gfx950): does vector work overlap the F16 matrix pipe?  (mfma_valu.hip answered it for the fp32 matrix instruction:
// no - fp32 MFMA and fp32 VALU share the FMA lanes and their times add, in one wave or in alternate waves.)
// v_mfma_f32_32x32x16_f16 alone, a VALU stream alone, MFMA waves beside VALU waves on the same SIMD, and both INTERLEAVED in one wave
// (one matrix instruction, then NV vector instructions, order pinned with sched_barrier) - for v_fma_f32, v_exp_f32 and v_cvt_pk_f16_f32.
//   hipcc --offload-arch=gfx950 -O3 scripts/microbench/mfma16_valu.hip -o /tmp/mfma16_valu && /tmp/mfma16_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// VOP: 0 v_fma_f32, 1 v_exp_f32, 2 v_cvt_pk_f16_f32 (+ v_fma_mix lo), 3 v_max_f32 / v_cndmask mix
template <int VOP>
__device__ __forceinline__ void valu_n(float (&v)[32], float pa, float pb, int n0, int n1) {
#pragma unroll
  for (int i = n0; i < n1; ++i) {
    const int j = i & 31;
    if (VOP == 0) v[j] = __builtin_fmaf(pa, pb, v[j]);
    if (VOP == 1) v[j] = __builtin_amdgcn_exp2f(v[j]);
    if (VOP == 2) {
      f16x2 h = {(_Float16)v[j], (_Float16)v[(j + 1) & 31]};
      v[j] += (float)h[0] * pa + (float)h[1];
    }
    if (VOP == 3) v[j] = fmaxf(v[j], v[(j + 7) & 31] + pa);
  }
}

// MODE 0: mfma only; 1: valu only; 2: alternate waves (even: mfma, odd: valu); 3: interleaved in one wave, NV vector instructions per matrix instruction
template <int MODE, int VOP, int NV>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float v[32];
  for (int i = 0; i < 32; ++i) v[i] = (float)i * 1e-3f + lane * 1e-5f;
  f16x8 fa, fb;
  for (int e = 0; e < 8; ++e) fa[e] = (_Float16)(a0 + 0.01f * e + lane * 1e-3f), fb[e] = (_Float16)(b0 - 0.02f * e);
  const float pa = a0 + lane * 1e-6f, pb = b0;
  const bool do_m = MODE == 0 || MODE == 3 || (MODE == 2 && (wave & 1) == 0);
  const bool do_v = MODE == 1 || MODE == 3 || (MODE == 2 && (wave & 1) == 1);
  for (int it = 0; it < iters; ++it) {
    if (MODE == 3) {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        acc[s & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc[s & 3], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        valu_n<VOP>(v, pa, pb, s * NV, s * NV + NV);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      if (do_m) {
#pragma unroll
        for (int s = 0; s < 16; ++s) acc[s & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc[s & 3], 0, 0, 0);
      }
      if (do_v) valu_n<VOP>(v, pa, pb, 0, 16 * NV);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 32; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int VOP, int NV> void run(const char* name, float* d, int blocks) {
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, VOP, NV>), dim3(blocks), dim3(256), 0, 0, d, 100, 1.0f, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, VOP, NV>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double waves = (double)blocks * 4;
  const double mwaves = MODE == 2 ? waves / 2 : (MODE == 0 || MODE == 3 ? waves : 0);
  const double vwaves = MODE == 2 ? waves / 2 : (MODE == 1 || MODE == 3 ? waves : 0);
  const double mf = mwaves * iters * 16.0 * 2.0 * 32 * 32 * 16;   // FLOP
  const double vi = vwaves * iters * 16.0 * NV;                    // source-level vector ops (wave instructions, roughly)
  printf("%-52s %8.3f ms  f16 mfma %7.1f TF   valu %7.2f Gop/s\n", name, ms, mf / ms / 1e9, vi / ms / 1e6);
}

template <int VOP, int NV> void suite(const char* vname, float* d) {
  char nm[96];
  const int blocks = 512;   // 2 workgroups of 4 waves per CU: 2 waves per SIMD
  snprintf(nm, sizeof nm, "%s x%d only", vname, NV); run<1, VOP, NV>(nm, d, blocks);
  snprintf(nm, sizeof nm, "mfma waves beside %s x%d waves (alternate)", vname, NV); run<2, VOP, NV>(nm, d, blocks);
  snprintf(nm, sizeof nm, "1 mfma : %d %s interleaved in one wave", NV, vname); run<3, VOP, NV>(nm, d, blocks);
}

int main() {
  float* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
  run<0, 0, 8>("f16 mfma only (2 waves per SIMD)", d, 512);
  run<0, 0, 8>("f16 mfma only (1 wave per SIMD)", d, 256);
  suite<0, 4>("v_fma_f32", d);
  suite<0, 8>("v_fma_f32", d);
  suite<0, 16>("v_fma_f32", d);
  suite<1, 2>("v_exp_f32", d);
  suite<1, 4>("v_exp_f32", d);
  suite<2, 4>("cvt_pk_f16 + mix", d);
  suite<3, 8>("v_max / add", d);
  return 0;
}
