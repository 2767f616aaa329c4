// Shared pieces of the three-product f16 GEMM kernels (gemm_split_f16.hip: the 4-wave tiles; gemm_split_ws.hip: the
// wave-specialised 192x256 tile).  Arithmetic and operand layout: see the header of gemm_split_f16.hip.
#pragma once
#include "common.hpp"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct SplitParams {
  const float* A;       // fp32 [M][lda], or packed planes [M][K/16][16 hi | 16 lo*2^11] f16 (APACK)
  const float* W;       // packed planes [N][K/16][16 hi | 16 lo] f16 of W[n] * 2^s(n)
  int wblk;             // W is in the BLOCKED layout [N/64][K/16][64 rows][16 hi | 16 lo] (pmce_gemm_pack_split_f16_blk): a tile's k-slice is
                        // contiguous 4 KB pieces instead of 64-byte pieces one weight row (K * 4 bytes) apart
  const float* wscale;  // [N]: 2^-s(n), one power of two per OUTPUT row of W (pmce_gemm_pack_split_f16)
  const float* bias;    // [N] or null
  const float* rscale;  // [M] or null: 2^e(m) per ROW of a packed A that was stored as A[m] * 2^-e(m) (pmce_split_rows_scaled_f16): the
                        // epilogue multiplies row m by it and adds the bias AFTER the scaling (RS instantiations)
  const float* R;       // residual [M][ldc] or null
  float* C;
  int M, N, K;
  unsigned lda, ldc;
  int ntm, ntn;
  int c_div;             // > 0: C row r lives at (r % c_div) * c_lo + (r / c_div) * c_hi (elements); never with R
  long long c_lo, c_hi;
  unsigned long long* clk;  // optional probe {shader clocks, 100 MHz ticks}: each workgroup's first wave adds its kernel residence (null = off)
  unsigned* oflow;  // device-visible word set to 1 when a result is not finite (an activation beyond f16's 65504, or fp32 overflow); may be null
  // LayerNorm epilogue (LNEP instantiation; N == 256 == the tile's width, so a workgroup owns whole rows): with x = the product + bias + R,
  //   y1 = ln1_w ? LN(x; ln1_w, ln1_b, ln1_eps) : x ;  out1 = y1 (fp32, if out1) ;  out2 = LN(y1; ln2_w, ln2_b, ln2_eps) pre-split (if out2)
  // - what pmce_ln_chain does to the product's result in a launch of its own.  C is not written.
  const float *ln1_w, *ln1_b, *ln2_w, *ln2_b;
  float ln1_eps, ln2_eps;
  float *out1, *out2;
};

// LDS-DMA: 64 lanes x 16 B from per-lane buffer offsets into LDS at M0 + lane*16 (see gemm_f32.hip)
__device__ __forceinline__ void sdma16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, int soff, unsigned lds_wave_base) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(rsrc), "s"(lds_wave_base), "s"(soff)
      : "memory");
}

// The same with the LDS destination as base + compile-time offset - what the k-loop issues (round 6): ONE base scalar per stage instead of one live
// scalar per instruction (the compiler had parked those in VGPR lanes and fetched each with a v_readlane per k-tile - vector instructions, which cost
// matrix time on this chip).
__device__ __forceinline__ void sdma16o(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, int soff, unsigned lds_base, int imm) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_add_u32 m0, %3, %5\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(rsrc), "s"(lds_base), "s"(soff), "n"(imm)
      : "memory", "scc");
}

// fp32 -> (hi, lo * 2^11) f16
__device__ __forceinline__ void split8(const f32x4& x0, const f32x4& x1, f16x8& hi, f16x8& lo) {
  const float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
  for (int e = 0; e < 8; ++e) hi[e] = (_Float16)v[e];
#pragma unroll
  for (int e = 0; e < 8; ++e) lo[e] = (_Float16)((v[e] - (float)hi[e]) * 2048.0f);
}

// 1 / x for x = 2^e, e in [-126, 126], exactly: the exponent field mirrored around the bias
__device__ __forceinline__ float pow2_recip(float x) { return __uint_as_float(0x7f000000u - __float_as_uint(x)); }

// exponent field all ones: infinity or NaN
__device__ __forceinline__ bool nonfinite(float v) { return __builtin_amdgcn_class(v, 0x207); }  // signalling / quiet NaN, -inf, +inf (0x203, until round 6, lacked the -inf bit)
__device__ __forceinline__ void report_nonfinite(unsigned* sink, bool lane_saw_one) {
  if (sink && __builtin_amdgcn_ballot_w64(lane_saw_one) != 0ull && (threadIdx.x & 63) == 0) *reinterpret_cast<volatile unsigned*>(sink) = 1u;
}

// One element of a PRE-SPLIT result: x -> (hi, lo * 2^11) f16 in one dword, exchanged with the neighbour lane (lane ^ 1) so that even lanes store
// {hi(n), hi(n + 1)} and odd lanes {lo(n - 1), lo(n)} - the layout [16 hi | 16 lo] of a 16-column group.  perm_sel = split_perm_sel(lane).
// Round 6: 20 instead of 25 vector instructions per element beside the GELU (v_pack_b32_f16, v_perm_b32, the class test on the f16 itself) -
// the epilogue's vector work costs the CU's other workgroup matrix time.  The same bits as the shift / mask form it replaces.
__device__ __forceinline__ unsigned split_perm_sel(int lane) { return (lane & 1) ? 0x03020706u : 0x05040100u; }
__device__ __forceinline__ unsigned split_pack_exchange(float x, unsigned perm_sel, bool& bad) {
  typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
  const _Float16 h = (_Float16)x;
  bad = bad || __builtin_amdgcn_classh(h, 0x207);  // inf / nan: also a finite x beyond f16's 65504 - the next product could not read it
  // (x - h) * 2^11 as ONE fused multiply-add on the f16 itself (v_fma_mix: no separate conversion and subtraction): x - h and the products by
  // 2^11 are exact in fp32, so the fused form has the bits of the three-instruction form
  const _Float16 l = (_Float16)fmaf((float)h, -2048.0f, x * 2048.0f);
  const unsigned w = __builtin_bit_cast(unsigned, h2_t{h, l});
  const unsigned nbr = (unsigned)__builtin_amdgcn_update_dpp(0, (int)w, 0xB1, 0xf, 0xf, true);  // lane ^ 1
  return __builtin_amdgcn_perm(nbr, w, perm_sel);
}

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}


// gemm_split_small.hip: the small-grid form (one 64 x 128 tile per workgroup, eight waves, at most one round of workgroups)
bool pmce_gemm_split_small_applies(int M, int N, int K, int act, bool apack, bool opack, bool res, bool rs);
int pmce_gemm_split_small_launch(SplitParams& p, int act, bool apack, bool opack, hipStream_t stream);
