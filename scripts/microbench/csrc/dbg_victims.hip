// Debugging aid (not part of the product path): self-checking bystander kernels.  Each wave recomputes one fixed function of
// its own registers over and over and counts the iterations whose result differs from the first - any count above zero means
// something outside the wave changed its arithmetic.  scripts/microbench/victims.py runs them next to a matrix-pipe kernel.
#include "common.hpp"

typedef float f32x2v __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void victim_kernel(unsigned* __restrict__ bad, int iters, float seed, const float* __restrict__ tab) {
  const int lane = threadIdx.x & 63;
  const float a = seed + 0.001f * (float)(threadIdx.x + 1), b = 1.0f + 0.0001f * (float)lane;
  unsigned mism = 0;
  float first0 = 0.f, first1 = 0.f, first2 = 0.f;
  for (int it = 0; it < iters; ++it) {
    float r0 = 0.f, r1 = 0.f, r2 = 0.f;
    float x = a, y = b;
    asm volatile("" : "+v"(x), "+v"(y));  // opaque: recomputed every iteration
    if (KIND == 0) {  // packed fp32 math
      f32x2v p = {x, y}, q = {y, x}, acc = {0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        acc = p * q + acc;
        p = p + f32x2v{0.25f, 0.5f};
        q = q * f32x2v{0.999f, 1.001f};
      }
      r0 = acc.x; r1 = acc.y;
    } else if (KIND == 1) {  // plain fp32 math
      float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        acc0 = fmaf(x, y, acc0);
        acc1 = fmaf(y, x + 1.f, acc1);
        x += 0.25f;
        y *= 0.999f;
      }
      r0 = acc0; r1 = acc1;
    } else if (KIND == 2) {  // three cross-lane butterfly sums issued together (ds_bpermute_b32, counted lgkmcnt waits)
      float s0 = x, s1 = y, s2 = x - y;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float t0 = __shfl_xor(s0, o, 64), t1 = __shfl_xor(s1, o, 64), t2 = __shfl_xor(s2, o, 64);
        s0 += t0; s1 += t1; s2 += t2;
      }
      r0 = s0; r1 = s1; r2 = s2;
    } else if (KIND == 3) {  // fp32 matrix pipe
      f32x16 c = {0};
#pragma unroll
      for (int k = 0; k < 8; ++k) c = __builtin_amdgcn_mfma_f32_32x32x2f32(x + (float)k, y, c, 0, 0, 0);
      r0 = c[0]; r1 = c[7]; r2 = c[15];
    } else if (KIND == 4) {  // DPP row operations (wave-local, no LDS unit)
      float s0 = x, s1 = y;
      s0 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s0), 0x111, 0xf, 0xf, true));  // row_shr:1
      s1 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s1), 0x112, 0xf, 0xf, true));  // row_shr:2
      s0 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s0), 0x114, 0xf, 0xf, true));
      r0 = s0; r1 = s1;
    } else if (KIND == 5) {  // one cross-lane sum at a time (ds_bpermute_b32, lgkmcnt(0) after each)
      float s0 = x;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float t0 = __shfl_xor(s0, o, 64);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        s0 += t0;
      }
      r0 = s0;
    }
    if (KIND == 6) {  // packed fp32 fma with the op_sel / op_sel_hi broadcast forms the lifter head uses
      f32x2v p = {x, y}, q = {y + 1.f, x + 2.f}, acc = {0.f, 0.f}, acc2 = {0.25f, 0.75f};
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        asm volatile("v_pk_fma_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(acc) : "v"(p), "v"(q));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc) : "v"(p), "v"(q));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc2) : "v"(q), "v"(acc));
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(acc2));
      }
      r0 = acc2.x; r1 = acc2.y; r2 = p.x + p.y;
    } else if (KIND == 7) {  // five 16-byte global loads in flight (table of known values), summed
      const float* t = tab + ((it * 977 + blockIdx.x * 131) & 4095) * 1024 + lane * 4;
      const f32x4 l0 = *reinterpret_cast<const f32x4*>(t), l1 = *reinterpret_cast<const f32x4*>(t + 256),
                  l2 = *reinterpret_cast<const f32x4*>(t + 512), l3 = *reinterpret_cast<const f32x4*>(t + 768),
                  l4 = *reinterpret_cast<const f32x4*>(t + 1024 * 17);
      // table[i] = (i % 8191) * 0.5: compare with the closed form
      auto want = [&](const float* q, int e) { return (float)((int)((q - tab) + e) % 8191) * 0.5f; };
      unsigned m = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        m += (l0[e] != want(t, e)) + (l1[e] != want(t + 256, e)) + (l2[e] != want(t + 512, e)) + (l3[e] != want(t + 768, e)) +
             (l4[e] != want(t + 1024 * 17, e));
      if (m) mism += 1;
      continue;
    }
    if (it == 0) {
      first0 = r0; first1 = r1; first2 = r2;
    } else {
      mism += (r0 != first0) + 16 * (r1 != first1) + 256 * (r2 != first2);
    }
  }
  if (mism) atomicAdd(bad + (mism & 15 ? 0 : 0), 1u), atomicAdd(bad + 1, mism & 15), atomicAdd(bad + 2, (mism >> 4) & 15), atomicAdd(bad + 3, mism >> 8);
}

extern "C" int pmce_dbg_victim(int kind, unsigned* bad4, int blocks, int iters, const float* tab, hipStream_t stream) {
  const dim3 g(blocks), b(256);
  switch (kind) {
    case 0: hipLaunchKernelGGL(victim_kernel<0>, g, b, 0, stream, bad4, iters, 0.5f, tab); break;
    case 1: hipLaunchKernelGGL(victim_kernel<1>, g, b, 0, stream, bad4, iters, 0.5f, tab); break;
    case 2: hipLaunchKernelGGL(victim_kernel<2>, g, b, 0, stream, bad4, iters, 0.5f, tab); break;
    case 3: hipLaunchKernelGGL(victim_kernel<3>, g, b, 0, stream, bad4, iters, 0.5f, tab); break;
    case 4: hipLaunchKernelGGL(victim_kernel<4>, g, b, 0, stream, bad4, iters, 0.5f, tab); break;
    case 5: hipLaunchKernelGGL(victim_kernel<5>, g, b, 0, stream, bad4, iters, 0.5f, tab); break;
    case 6: hipLaunchKernelGGL(victim_kernel<6>, g, b, 0, stream, bad4, iters, 0.5f, tab); break;
    default: hipLaunchKernelGGL(victim_kernel<7>, g, b, 0, stream, bad4, iters, 0.5f, tab); break;
  }
  return pmce_check_launch("dbg_victim");
}

// ---- matrix-pipe spinners: nothing but one MFMA shape on register operands (no memory, no LDS), to find which shapes disturb
// a bystander ----
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ __launch_bounds__(256) void mfma_spin_kernel(float* __restrict__ sink, int iters) {
  const float s = 1.0f + 0.001f * (float)threadIdx.x;
  f32x16 c0 = {0}, c1 = {0};
  f4 d0 = {0, 0, 0, 0}, d1 = {0, 0, 0, 0};
  h8 a, b;
  b8 ab, bb;
  h4 a4, b4;
#pragma unroll
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(s + e); b[e] = (_Float16)(0.5f * s - e); ab[e] = (__bf16)(s + e); bb[e] = (__bf16)(0.5f * s - e); }
#pragma unroll
  for (int e = 0; e < 4; ++e) { a4[e] = (_Float16)(s + e); b4[e] = (_Float16)(0.5f * s - e); }
  const long al = __builtin_bit_cast(long, a4), bl = __builtin_bit_cast(long, b4);
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) { c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c1, 0, 0, 0); }
    if (KIND == 1) { d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, d1, 0, 0, 0); }
    if (KIND == 2) { c0 = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x8f16(b4, a4, c1, 0, 0, 0); }
    if (KIND == 3) { c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bb, ab, c1, 0, 0, 0); }
    if (KIND == 4) { c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(s, 0.5f, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(0.5f, s, c1, 0, 0, 0); }
    if (KIND == 5) { c0 = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(al, bl, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(bl, al, c1, 0, 0, 0); }
    if (KIND == 6) { d0 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_16x16x16f16(b4, a4, d1, 0, 0, 0); }
  }
  float r = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) r += c0[e] + c1[e];
  r += d0[0] + d0[1] + d0[2] + d0[3] + d1[0] + d1[1] + d1[2] + d1[3];
  if (r == 12345.678f) sink[threadIdx.x] = r;  // keep the chain alive
}
extern "C" int pmce_dbg_mfma_spin(int kind, float* sink, int blocks, int iters, hipStream_t stream) {
  const dim3 g(blocks), b(256);
  switch (kind) {
    case 0: hipLaunchKernelGGL(mfma_spin_kernel<0>, g, b, 0, stream, sink, iters); break;
    case 1: hipLaunchKernelGGL(mfma_spin_kernel<1>, g, b, 0, stream, sink, iters); break;
    case 2: hipLaunchKernelGGL(mfma_spin_kernel<2>, g, b, 0, stream, sink, iters); break;
    case 3: hipLaunchKernelGGL(mfma_spin_kernel<3>, g, b, 0, stream, sink, iters); break;
    case 4: hipLaunchKernelGGL(mfma_spin_kernel<4>, g, b, 0, stream, sink, iters); break;
    case 5: hipLaunchKernelGGL(mfma_spin_kernel<5>, g, b, 0, stream, sink, iters); break;
    default: hipLaunchKernelGGL(mfma_spin_kernel<6>, g, b, 0, stream, sink, iters); break;
  }
  return pmce_check_launch("dbg_mfma_spin");
}

// One v_mfma_f32_32x32x16_f16 with A = a, B = b in every element: out[0] = the (uniform) result = 16 a b when the matrix pipe
// reads subnormal f16 inputs as they are, out[1] = a after the conversion to f16.  vertex_sa's f16 form stores the lo halves of
// small K / V elements as subnormals (coevo.hip); the test pins the hardware behaviour it relies on.
typedef _Float16 dbg_f16x8 __attribute__((ext_vector_type(8)));
__global__ void mfma_subnormal_kernel(float* out, float a, float b) {
  dbg_f16x8 A, B;
  for (int e = 0; e < 8; ++e) {
    A[e] = (_Float16)a;
    B[e] = (_Float16)b;
  }
  f32x16 C;
  for (int r = 0; r < 16; ++r) C[r] = 0.f;
  C = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0);
  if (threadIdx.x == 0) {
    out[0] = C[0];
    out[1] = (float)A[0];
  }
}
extern "C" int pmce_dbg_mfma_subnormal(float a, float b, float* out2, hipStream_t stream) {
  PMCE_REQUIRE(out2, "dbg_mfma_subnormal: null pointer");
  hipLaunchKernelGGL(mfma_subnormal_kernel, dim3(1), dim3(64), 0, stream, out2, a, b);
  return pmce_check_launch("dbg_mfma_subnormal");
}
