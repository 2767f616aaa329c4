// The three-product f16 GEMM of gemm_split_f16.hip for SMALL GRIDS (round 5): ONE tile per workgroup, at most one round of workgroups on the
// chip - the shape of every product of a single-clip forward (M = 272 tokens, 16 frames, 1 row) and of the small-M products of any batch (AdaLN
// parameters, the final product, the GRU projections of small batches).
//
// Such a launch is bound by no throughput: measured (scripts/microbench/chain_latency.hip, dma_patterns.hip, small_m_tiles.py) a dependent
// v_mfma_f32_32x32x16_f16 issues after 44 cycles, an LDS-DMA instruction costs a lone workgroup's CU 16 cycles whatever its row pattern, an L2
// round trip is 300 cycles - and the persistent kernel's 64 x 128 tile took 1,510 cycles per pair of k-tiles: four waves, ONE per SIMD, each
// walking its ~200 instructions per trip (tile-stream bookkeeping, six DMA issues, twelve fragment reads, two chains of three dependent matrix
// instructions) with nothing on the SIMD to issue in the gaps; and the weights of a single-clip forward come from HBM (0.6 GB pass through the
// caches per forward), where what a CU receives is its requests in flight divided by a microsecond.
// Here a SIMD always holds TWO waves with ONE 32 x 32 accumulator each (a wave's dependent chain overlaps its partner's): a 64 x 128 tile of
// eight waves, or - when 256 of them cover the product, or 512 for products of at most 32 rows (pure weight streaming: twice the workgroups
// keep twice the requests in flight) - a 64 x 64 tile of four waves with two workgroups per CU.  No tile stream (the loop is wait - barrier -
// issue - multiply), bias / scales / residual requested before the first k-tile instead of after the last, a ring of six 24 KB or four 16 KB
// stages of two k-tiles each.
// Arithmetic: the same k-tiles in the same order into one fp32 accumulator per element, the same epilogue expressions - results are
// bit-identical to the persistent kernel's (tests: batch invariance B = 4 against B = 64; test_gemm_split_small_grid_equals_persistent).
#include <atomic>

#include "gemm_split_common.hpp"

// NWN = 4: 64 x 128 tile, eight waves (2 x 4), six stages of 24 KB, one workgroup per CU.  NWN = 2: 64 x 64 tile, four waves (2 x 2), four stages
// of 16 KB, two workgroups per CU - for products of at most 64 rows, which are pure weight streaming (the final product of a single clip reads
// 278 MB): twice the workgroups keep twice the requests in flight towards HBM.
template <int NWN>
struct SmallCfg {
  static constexpr int BM = 64, BN = 32 * NWN, NW = 2 * NWN;
  static constexpr int SUBF = (BM + BN) * 16;   // floats of one 16-wide k-tile in LDS: BM + BN rows of 64 B
  static constexpr int SF = 2 * SUBF;           // a stage = two k-tiles = 24 / 16 KB
  static constexpr int NS = NWN == 4 ? 6 : 4;   // 144 KB / 64 KB
  static constexpr int LDS = NS * SF * 4;
  static constexpr int GPS = (BM + BN) / 16;    // 16-row groups (DMA instructions) per k-tile: 12 / 8
  static constexpr int DPW = 2 * GPS / NW;      // per wave and stage: 3 / 4
};

template <int NWN, int ACT, bool RES, bool APACK, bool OPACK, bool RS>
__global__ __launch_bounds__(128 * NWN) void gemm_split_small_kernel(SplitParams p) {
  static_assert(!RS || (APACK && !OPACK && !RES && ACT == 0), "a row-scaled A is a packed A; no packed result, residual or activation");
  using Cfg = SmallCfg<NWN>;
  constexpr int S_BM = Cfg::BM, S_BN = Cfg::BN, S_SUBF = Cfg::SUBF, S_SF = Cfg::SF, S_NS = Cfg::NS, GPS = Cfg::GPS, DPW = Cfg::DPW, NW = Cfg::NW;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n0 = lane & 31, hb = lane >> 5;
  const int wm = wave & 1, wn = wave >> 1;

  // tile of this workgroup: XCD-local chunks of the tile order, row tiles fastest (the row tiles of one W panel share an L2)
  const int nblk = p.ntm * p.ntn;
  const int xcd = blockIdx.x & 7, bx = blockIdx.x >> 3;
  const int cq = nblk >> 3, cr = nblk & 7;
  const int chunk_start = xcd < cr ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq;
  if (bx >= cq + (xcd < cr ? 1 : 0)) return;
  const int t = chunk_start + bx;
  const int m_base = (t % p.ntm) * S_BM, n_base = (t / p.ntm) * S_BN;

  // ---- everything the epilogue needs from memory is requested now (in-order completion: it lands before the first k-tile) ----
  // (unconditional loads from clamped addresses: the compiler can count them, and its own wait for the bias does not drain the ring)
  const int n = n_base + wn * 32 + n0;
  const bool n_ok = n < p.N;
  const int nc = min(n, p.N - 1);
  const float w_down = p.wscale[nc];
  const float bias_v = (p.bias ? p.bias : p.wscale)[nc];
  const float bias_n = p.bias ? bias_v : 0.f;
  float pre[16];  // RES: the residual of this lane's 16 elements; RS: 2^e of its 16 rows
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int mc = min(m_base + wm * 32 + 4 * hb + (r & 3) + 8 * (r >> 2), p.M - 1);
    if constexpr (RES) pre[r] = p.R[(size_t)mc * p.ldc + nc];
    else if constexpr (RS) pre[r] = p.rscale[mc];
    else pre[r] = 0.f;
  }
  asm volatile("" ::: "memory");

  // ---- DMA side: 2 GPS instructions of 16 rows x 64 B per stage, DPW per wave: instruction j = wave + NW q is row group g = j % GPS (0-3: A,
  // the rest: W) of the stage's k-tile j / GPS; lane L -> row 16 g + (L >> 2), PHYSICAL chunk L & 3 = logical chunk (L & 3) ^ ((L >> 4) & 3) ----
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, 0xffffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.W), 0, 0xffffffff, 0x00020000);
  const int drow = lane >> 2;
  const unsigned dchunk = (unsigned)(((lane & 3) ^ ((lane >> 4) & 3)) * 4);  // floats
  const int kstep_w = p.wblk ? 4096 : 64;  // bytes from one k-tile of a W row (block) to the next
  unsigned doff[DPW];
  int dlds[DPW], dks[DPW], dk0[DPW];
  bool d_is_a[DPW];
#pragma unroll
  for (int q = 0; q < DPW; ++q) {
    const int j = wave + NW * q, sub = j / GPS, g = j % GPS;
    d_is_a[q] = g < 4;
    if (g < 4)
      doff[q] = ((unsigned)min(m_base + 16 * g + drow, p.M - 1) * p.lda + dchunk) * 4u;
    else {
      const unsigned r = (unsigned)min(n_base + 16 * (g - 4) + drow, p.N - 1);
      doff[q] = p.wblk ? ((r >> 6) * (unsigned)(p.K / 16) * 1024u + (r & 63u) * 16u + dchunk) * 4u : (r * (unsigned)p.K + dchunk) * 4u;
    }
    dlds[q] = sub * (S_SUBF * 4) + g * 1024;
    dks[q] = g < 4 ? 64 : kstep_w;
    dk0[q] = sub * dks[q];
  }
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)lds);
  int i_s = 0, i_slot = 0;  // next stage to request (k-tiles 2 s, 2 s + 1) and its ring slot
  auto issue_next = [&]() {
#pragma unroll
    for (int q = 0; q < DPW; ++q)
      sdma16(d_is_a[q] ? rsrc_a : rsrc_w, doff[q], dk0[q] + 2 * i_s * dks[q], lds0 + i_slot * (S_SF * 4) + dlds[q]);
    ++i_s;
    i_slot = i_slot + 1 == S_NS ? 0 : i_slot + 1;
  };
  const int nk = p.K / 32;
#pragma unroll
  for (int s = 0; s < S_NS - 1; ++s)
    if (s < nk) issue_next();

  const int swz = (n0 >> 2) & 3;
  const int a_row = (wm * 32 + n0) * 16, w_row = S_BM * 16 + (wn * 32 + n0) * 16;  // floats inside a k-tile
  const int ca0 = 4 * ((2 * hb) ^ swz), ca1 = 4 * ((2 * hb + 1) ^ swz);            // fp32 A: k = 8 hb + [0,4), + [4,8)
  const int ch = 4 * (hb ^ swz), cl = 4 * ((2 + hb) ^ swz);                        // packed: hi / lo plane, k = 8 hb + [0,8)

  f32x16 acc;
  {  // the bias (scaled like its row of W) is the accumulator's initial value
    const float bv = (p.bias && !RS) ? bias_n * pow2_recip(w_down) : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bv;
  }
  int slot = 0;
  for (int it = 0; it < nk; ++it) {
    // stage `it` has landed when at most the younger stages' DMAs (DPW per wave each) are in flight
    const int younger = min(nk - 1 - it, S_NS - 2);
    if (S_NS >= 6 && younger >= 4) wait_vm<4 * DPW>();
    else if (S_NS >= 5 && younger == 3) wait_vm<3 * DPW>();
    else if (younger == 2) wait_vm<2 * DPW>();
    else if (younger == 1) wait_vm<DPW>();
    else wait_vm<0>();
    __syncthreads();  // every wave's part of stage `it` is in LDS; every wave is done reading stage it - 1
    if (i_s < nk) issue_next();
    const float* st = lds + slot * S_SF;
    slot = slot + 1 == S_NS ? 0 : slot + 1;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const float* sA = st + sub * S_SUBF;
      f16x8 ahi, alo;
      if constexpr (APACK) {
        ahi = *reinterpret_cast<const f16x8*>(sA + a_row + ch);
        alo = *reinterpret_cast<const f16x8*>(sA + a_row + cl);
      } else {
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(sA + a_row + ca0);
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(sA + a_row + ca1);
        split8(x0, x1, ahi, alo);
      }
      const f16x8 whi = *reinterpret_cast<const f16x8*>(sA + w_row + ch);
      const f16x8 wlo = *reinterpret_cast<const f16x8*>(sA + w_row + cl);
      const f16x8 wh2 = whi * (_Float16)0.00048828125f;  // 2^-11: undoes the scale of alo
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, whi, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, wlo, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, wh2, acc, 0, 0, 0);
    }
  }

  // ---- epilogue: the expressions of gemm_split_kernel's, element for element ----
  bool bad = false;
  const int m_lane = m_base + wm * 32 + 4 * hb;  // + (r & 3) + 8 (r >> 2)
  if constexpr (RS) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int mm = m_lane + (r & 3) + 8 * (r >> 2);
      if (mm < p.M) {
        float* crow = p.c_div > 0 ? p.C + (long long)(mm % p.c_div) * p.c_lo + (long long)(mm / p.c_div) * p.c_hi : p.C + (size_t)mm * p.ldc;
        const float v = fmaf(acc[r] * w_down, pre[r], bias_n);
        bad = bad || nonfinite(v);
        if (n_ok) __builtin_nontemporal_store(v, crow + n);
      }
    }
  } else if constexpr (OPACK) {
    // pre-split result [row][K/16][16 hi | 16 lo*2^11] f16 in the bytes of the fp32 row: adjacent lanes pair up (see gemm_split_kernel)
    const bool odd = lane & 1;
    const unsigned psel = split_perm_sel(lane);
    const int colf = (n0 >> 4) * 32 + (odd ? 16 + ((n0 - 1) & 15) : (n0 & 15));
    const int cb = n_base + wn * 32;
    if (cb < p.N) {  // (N % 32 == 0: a 32-column block is all in or all out)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        f32x2 v = {acc[r] * w_down, acc[r + 1] * w_down};
        if (ACT == 1) v = gelu_erf2(v);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const unsigned outw = split_pack_exchange(pinned(e ? v.y : v.x), psel, bad);   // (gemm_split_common.hpp: the persistent kernel's, bit for bit)
          const int mm = m_lane + ((r + e) & 3) + 8 * ((r + e) >> 2);
          if (mm < p.M) reinterpret_cast<unsigned*>(p.C + (size_t)mm * p.ldc + cb)[colf >> 1] = outw;
        }
      }
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      f32x2 v = {acc[r] * w_down, acc[r + 1] * w_down};
      if (ACT == 1) v = gelu_erf2(v);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int mm = m_lane + ((r + e) & 3) + 8 * ((r + e) >> 2);
        if (n_ok && mm < p.M) {
          float o = e ? v.y : v.x;
          if (RES) o += pre[r + e];
          bad = bad || nonfinite(o);
          float* crow = p.c_div > 0 ? p.C + (long long)(mm % p.c_div) * p.c_lo + (long long)(mm / p.c_div) * p.c_hi : p.C + (size_t)mm * p.ldc;
          crow[n] = o;
        }
      }
    }
  }
  report_nonfinite(p.oflow, bad);
}

template <int NWN, int ACT, bool RES, bool APACK, bool OPACK, bool RS>
static int launch_small(const SplitParams& p, hipStream_t stream) {
  static std::atomic<unsigned long long> done{0};
  constexpr int LDS = SmallCfg<NWN>::LDS;
  PMCE_TRY(pmce_opt_in_lds(reinterpret_cast<const void*>(&gemm_split_small_kernel<NWN, ACT, RES, APACK, OPACK, RS>), LDS, done, "gemm_split_small"));
  const int grid = ((p.ntm * p.ntn + 7) / 8) * 8;
  hipLaunchKernelGGL((gemm_split_small_kernel<NWN, ACT, RES, APACK, OPACK, RS>), dim3(grid), dim3(128 * NWN), LDS, stream, p);
  return PMCE_OK;
}

// Which tile?  64 x 64 (two workgroups per CU) when one round of at most 256 such tiles covers the product, or - products of at most 32 rows:
// pure weight streaming - at most 512 of them; else 64 x 128 if one round of at most 256 covers it; else none (0).  Measured, same box
// (profiles/r05_e_small_grid_gemm_ab.txt): 64 x 64 for every product of a single-clip forward -3.7 % per forward at C = 512, the AdaLN product
// 49 -> 37 us, the final product 82 -> 76 us up to B = 8 (at B = 64 its 323 tiles of 64 x 64 are slower than 162 of 64 x 128: hence the 32 rows).
static int small_nwn_for(int M, int N) {
  const long long t64 = (long long)((M + 63) / 64) * ((N + 63) / 64), t128 = (long long)((M + 63) / 64) * ((N + 127) / 128);
  if (t64 <= 256 || (M <= 32 && t64 <= 512)) return 2;
  return t128 <= 256 ? 4 : 0;
}

// Is there a small-grid form for this product?  (K in whole pairs of k-tiles; one round of workgroups; the operand / epilogue combinations
// the model's small products use.)
bool pmce_gemm_split_small_applies(int M, int N, int K, int act, bool apack, bool opack, bool res, bool rs) {
  if (K % 32 != 0 || K < 64) return false;
  if (small_nwn_for(M, N) == 0) return false;
  if (rs) return K >= 128;  // (the row-scaled form's documented limit, enforced by gemm_split_any before either kernel is chosen)
  if (opack) return act == 1;
  if (apack) return act == 0;
  return act == 0 && !res;
}

template <int NWN>
static int launch_small_cfg(SplitParams& p, bool apack, bool opack, hipStream_t stream) {
  p.ntm = (p.M + SmallCfg<NWN>::BM - 1) / SmallCfg<NWN>::BM;
  p.ntn = (p.N + SmallCfg<NWN>::BN - 1) / SmallCfg<NWN>::BN;
  const bool res = p.R != nullptr;
  if (p.rscale) return launch_small<NWN, 0, false, true, false, true>(p, stream);
  if (opack) return launch_small<NWN, 1, false, true, true, false>(p, stream);
  if (apack) return res ? launch_small<NWN, 0, true, true, false, false>(p, stream) : launch_small<NWN, 0, false, true, false, false>(p, stream);
  return launch_small<NWN, 0, false, false, false, false>(p, stream);
}

int pmce_gemm_split_small_launch(SplitParams& p, int act, bool apack, bool opack, hipStream_t stream) {
  (void)act;
  return small_nwn_for(p.M, p.N) == 2 ? launch_small_cfg<2>(p, apack, opack, stream) : launch_small_cfg<4>(p, apack, opack, stream);
}
