// The three-product f16 GEMM of gemm_split_f16.hip on the OTHER full-rate f16 matrix shape, v_mfma_f32_16x16x32_f16.
//
// Why.  MI355X runs these kernels at its power limit, and the two shapes do not cost the same energy: from register operands
// holding random f16 bit patterns, 32x32x16 sustains 1.53 PFLOP/s at 1.48 GHz and 16x16x32 1.76 PFLOP/s at 1.72 GHz, both with
// the pipe 98 % occupied (scripts/microbench/mfma_peak.hip, profiles/r03_d_mfma_peak_register_operands.txt); the GEMM's
// fragment-read + matrix-instruction loop: 54 -> 61 % of the nominal peak (scripts/microbench/lds_read_bw.hip).
//
// Same operands, same ring, same arithmetic per element as gemm_split_kernel (pre-split A only: the lifter's 24 block products):
//     C = 2^-s ( sum ahi whi + ahi wlo + alo (whi 2^-11) ),     hi / lo planes as [row][K/16][16 hi | 16 lo] f16
// but the K = 32 of the instruction is filled by CONCATENATING PRODUCTS of one 16-wide k-tile instead of two k-tiles, so that the
// ring (one k-tile per stage, one barrier per k-tile) stays as it is.  Lane l of a 16x16x32 operand holds row l & 15 and the 8
// k-slots of group g = l >> 4:
//     I1, every k-tile:   A slots = [ahi(k 0-7) | ahi(k 8-15) | ahi(k 0-7) | ahi(k 8-15)],  B slots = [whi | whi | wlo | wlo]
//                         -> ahi whi + ahi wlo of the k-tile in ONE instruction;
//     I3, every 2nd one:  A slots = [alo(t0) (k 0-7 | 8-15) | alo(t1) (...)],  B slots = [wh2(t0) | wh2(t1)], wh2 = whi 2^-11
//                         -> the alo whi product of TWO k-tiles; its operands of the even k-tile wait in the g < 2 lanes' registers.
// 1.5 instructions of 16 cycles per 16x16 block and k-tile = the 3 of 32 cycles per 32x32 block of the other kernel.  Fragment
// reads per k-tile and wave (64 x 128 wave tile): 12 full ds_read_b128 (I1) + on average 8 half-masked ones (alo; whi of the odd
// k-tile for the g >= 2 lanes) against 12 - the price of not holding two k-tiles in LDS.
// The 16-byte chunks of a stage row are XOR-swizzled with (-(row >> 2)) & 3 here (DMA side and read side): with 16 rows x 2
// k-groups per half wave the service groups of ds_read_b128 need that permutation to fall on 16 different slots.
// (Unlike the default kernel, whose k order is the same for every tile shape, the 64 x 64 wave tile adds a pair's I3 one k-tile later
// than the 64 x 128 one: results of this kernel may differ in the last bit between batch sizes that select different tiles - one
// more reason it is an opt-in.)
// Accumulators: 16x16 blocks, lane l holds rows 4 (l >> 4) + r of ONE column; the W rows are permuted on their way into LDS so that
// the two blocks of a 32-column group give a lane the ADJACENT columns 2 (l & 15) and + 1: the epilogue moves 8 bytes per lane,
// 128 contiguous bytes per 16 lanes (with plain 16-column blocks it moved 64-byte half lines and the residual products lost 40 %).
#include <atomic>

#include "gemm_split_common.hpp"

namespace {

typedef float f32x4m __attribute__((ext_vector_type(4)));

template <int TM, int TN>
struct M16Cfg {
  static constexpr int BM = 64 * TM, BN = 64 * TN;
  static constexpr int STAGE_FLOATS = (BM + BN) * 16;
  static constexpr int NS = STAGE_FLOATS * 4 * 4 <= 64 * 1024 ? 4 : 3;
  static constexpr int LDS_BYTES = NS * STAGE_FLOATS * 4 + 4 * 1024;
  static constexpr int DPW = (BM + BN) / 64;
};

template <int TM, int TN, int ACT, bool RES, bool OPACK>
__global__ __launch_bounds__(256, 2) void gemm_split_m16_kernel(SplitParams p) {
  using Cfg = M16Cfg<TM, TN>;
  constexpr int BM = Cfg::BM, BN = Cfg::BN, WM = 32 * TM, WN = 32 * TN;
  constexpr int NS = Cfg::NS, SF = Cfg::STAGE_FLOATS, DPW = Cfg::DPW;
  constexpr int GA = BM / 16;
  constexpr int RB = 2 * TM, CB = 2 * TN;  // 16-row / 16-column blocks of the wave tile
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  const int wm = wave & 1, wn = wave >> 1;

  // ---- persistent workgroups on an XCD-local chunk of the grouped tile order (as gemm_split_kernel) ----
  const int nblk = p.ntm * p.ntn;
  const int xcd = blockIdx.x & 7, bx = blockIdx.x >> 3, gx = gridDim.x >> 3;
  const int cq = nblk >> 3, cr = nblk & 7;
  const int chunk_start = xcd < cr ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq;
  const int chunk_len = cq + (xcd < cr ? 1 : 0);
  if (bx >= chunk_len) return;
  constexpr int GROUP_M = 8;
  const int per_group = GROUP_M * p.ntn;
  auto tile_coords = [&](int bid, int& mb, int& nb) {
    const int group = bid / per_group;
    const int first_m = group * GROUP_M;
    const int gsz = min(p.ntm - first_m, GROUP_M);
    mb = (first_m + (bid % per_group) % gsz) * BM;
    nb = ((bid % per_group) / gsz) * BN;
  };
  const bool probe = p.clk != nullptr && tid == 0;
  const long long pc0 = probe ? (long long)__builtin_readcyclecounter() : 0, pw0 = probe ? (long long)wall_clock64() : 0;
  const int my_tiles = (chunk_len - bx + gx - 1) / gx;
  const int nk = p.K / 16;  // even (the launcher requires K % 32 == 0)
  const int total = my_tiles * nk;
  if (p.skew > 0 && gridDim.x >= 512 && bx >= (gx >> 1)) {
    for (int i = 0; i < p.skew; ++i) __builtin_amdgcn_s_sleep(64);
  }

  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, 0xffffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.W), 0, 0xffffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.bias ? p.N * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wscale), 0, p.N * 4, 0x00020000);

  // ---- DMA side: lane L -> row 16 g' + (L >> 2) of a stage, PHYSICAL chunk L & 3 = logical chunk (L & 3) ^ ((-(row >> 2)) & 3) ----
  const int drow = lane >> 2;
  const unsigned dchunk = (unsigned)(((lane & 3) ^ ((0 - (lane >> 4)) & 3)) * 4);  // floats
  unsigned doff[DPW];
  auto set_ptrs = [&](int mb, int nb) {
#pragma unroll
    for (int q = 0; q < DPW; ++q) {
      const int gg = wave + 4 * q;
      if (gg < GA)
        doff[q] = ((unsigned)min(mb + 16 * gg + drow, p.M - 1) * p.lda + dchunk) * 4u;
      else {
        // LDS row rho = 16 (gg - GA) + drow of the W part holds W row sigma(rho) = 32 (rho >> 5) + 2 (rho & 15) + ((rho >> 4) & 1): the
        // two 16-row blocks of a 32-row group take the even and the odd rows, so that lane r16 of blocks 2q and 2q + 1 owns the ADJACENT
        // output columns 32 q + 2 r16 and + 1 - one 8-byte access per lane, 128 contiguous bytes per 16 lanes, in the epilogue
        const int blk = gg - GA;
        const int wrow = nb + 32 * (blk >> 1) + 2 * drow + (blk & 1);
        doff[q] = ((unsigned)min(wrow, p.N - 1) * (unsigned)p.K + dchunk) * 4u;
      }
    }
  };
  const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)lds;
  const unsigned lds_wave = __builtin_amdgcn_readfirstlane(lds0 + wave * 1024);
  auto issue = [&](int kt, int stage) {
    const int ko = kt * 64;  // bytes
#pragma unroll
    for (int q = 0; q < DPW; ++q)
      sdma16((wave + 4 * q) < GA ? rsrc_a : rsrc_w, doff[q], ko, lds_wave + stage * (SF * 4) + q * 4096);
  };
  int i_li = bx, i_kt = 0, i_stage = 0, issued = 0, i_nb = 0, i_par = 0;
  {
    int mb;
    tile_coords(chunk_start + i_li, mb, i_nb);
    set_ptrs(mb, i_nb);
  }
  auto issue_next = [&]() {
    if (i_kt == 0) {
      if (p.bias && wave == 0) sdma16(rsrc_b, (unsigned)lane * 16u, i_nb * 4, lds0 + NS * SF * 4 + i_par * 2048);
      if (wave == 1) sdma16(rsrc_s, (unsigned)lane * 16u, i_nb * 4, lds0 + NS * SF * 4 + i_par * 2048 + 1024);
      i_par ^= 1;
    }
    issue(i_kt, i_stage);
    ++issued;
    i_stage = i_stage + 1 == NS ? 0 : i_stage + 1;
    if (++i_kt == nk) {
      i_kt = 0;
      i_li += gx;
      if (i_li < chunk_len) {
        int mb;
        tile_coords(chunk_start + i_li, mb, i_nb);
        set_ptrs(mb, i_nb);
      }
    }
  };
#pragma unroll
  for (int q = 0; q < NS - 1; ++q)
    if (issued < total) issue_next();

  // ---- read side: row r16 of a 16-row block; swizzle of that row; chunk offsets (floats) of this lane's fragments ----
  const int swz = (0 - (r16 >> 2)) & 3;
  const int a_row = (wm * WM + r16) * 16, w_row = BM * 16 + (wn * WN + r16) * 16;  // floats inside a stage (+ 256 per 16-row block)
  const int c_ahi = 4 * ((g & 1) ^ swz);        // I1 A: hi plane, k = 8 (g & 1) + [0, 8)   (lanes g and g + 2 read the same 16 bytes)
  const int c_alo = 4 * ((2 + (g & 1)) ^ swz);  // I3 A: lo plane
  const int c_w1 = 4 * (g ^ swz);               // I1 B: g < 2 the hi plane, g >= 2 the lo plane
  const int c_whi = 4 * ((g & 1) ^ swz);        // I3 B of the g >= 2 lanes: the hi plane again
  const _Float16 kDown = (_Float16)0.00048828125f;  // 2^-11: undoes the scale of alo

  float w_down[CB];  // 2^-s of this lane's column in each 16-column block
  f32x4m acc[RB][CB];
  f16x8 a3[RB], b3[CB];  // I3 operands: g < 2 lanes hold the even k-tile's, g >= 2 lanes the odd one's
  // DEFER (the 64 x 64 wave tile, which has the registers for it): a pair's I3 is issued at the START of the next pair's even
  // k-tile, where it depends on no fresh LDS read and runs under that k-tile's fragment reads; issued behind the odd k-tile's reads
  // it leaves the pipe idle for an LDS round trip per pair (PMC: 16.5 % more kernel cycles than the 32x32x16 kernel for the same
  // matrix-busy cycles).  One code path for every pair: a tile's first pair issues the "previous" I3 on ZERO operands (a3 / b3 are
  // cleared after every tile: 32 extra instructions per tile, 2 % at K = 512); the last pair's I3 goes out before the epilogue.
  constexpr bool DEFER = CB <= 4;
  auto clear_i3 = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) a3[i][e] = (_Float16)0.f;
#pragma unroll
    for (int j = 0; j < CB; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) b3[j][e] = (_Float16)0.f;
  };
  if constexpr (DEFER) clear_i3();
  int li = bx, kt = 0, stage = 0, c_par = 0;
  int m_base, n_base;
  tile_coords(chunk_start + li, m_base, n_base);

  // one ring step: k-tile `it` has landed (counted wait, barrier), the DMA of k-tile it + NS - 1 goes out; returns the stage to read
  auto ring_step = [&](int it) __attribute__((always_inline)) -> const float* {
    const int younger = issued - it - 1;
    if (NS == 4 && younger >= 2) wait_vm<2 * DPW>();
    else if (younger >= 1) wait_vm<DPW>();
    else wait_vm<0>();
    __syncthreads();
    if (issued < total) issue_next();
    const float* sA = lds + stage * SF;
    stage = stage + 1 == NS ? 0 : stage + 1;
    return sA;
  };
  auto read_i1 = [&](const float* sA, f16x8 (&a1)[RB], f16x8 (&b1)[CB]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < RB; ++i) a1[i] = *reinterpret_cast<const f16x8*>(sA + a_row + i * 256 + c_ahi);
#pragma unroll
    for (int j = 0; j < CB; ++j) b1[j] = *reinterpret_cast<const f16x8*>(sA + w_row + j * 256 + c_w1);
  };
  auto issue_i1 = [&](const f16x8 (&a1)[RB], const f16x8 (&b1)[CB]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int j = 0; j < CB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[i], b1[j], acc[i][j], 0, 0, 0);
  };

  // The k-loop runs in PAIRS of k-tiles (nk is even, so a pair never straddles two tiles).  Order inside a pair, chosen for the
  // register file (a1 / b1 and a3 / b3 are never alive together: 48 + 128 accumulator registers instead of 96 + 128):
  //   even k-tile:  read a1, b1 ; I1 ; then the g < 2 lanes keep  a3 = alo,  b3 = b1 2^-11  (b1 is whi there)
  //   odd k-tile:   the g >= 2 lanes read their a3 = alo, b3 = whi 2^-11 ; I3 ; only then read a1, b1 ; I1
  for (int it = 0; it < total; it += 2) {
    {
      const float* sA = ring_step(it);
      if (kt == 0) {  // the bias slice (scaled like its row of W) is the accumulators' initial value
        const float* sB = lds + NS * SF + c_par * 512 + wn * WN + 2 * r16;  // block j, lane r16: column 32 (j >> 1) + 2 r16 + (j & 1)
        c_par ^= 1;
#pragma unroll
        for (int j = 0; j < CB; ++j) {
          w_down[j] = sB[256 + 32 * (j >> 1) + (j & 1)];
          const float bv = p.bias ? sB[32 * (j >> 1) + (j & 1)] * pow2_recip(w_down[j]) : 0.f;
#pragma unroll
          for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = bv;
        }
      }
      f16x8 a1[RB], b1[CB];
      read_i1(sA, a1, b1);
      if constexpr (DEFER) {
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
          for (int j = 0; j < CB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a3[i], b3[j], acc[i][j], 0, 0, 0);
      }
      issue_i1(a1, b1);
      // (every lane writes, although only the g < 2 lanes' values are used - the g >= 2 lanes overwrite theirs in the odd k-tile:
      // a lane-masked write would keep the 48 registers of a3 / b3 alive through the epilogue of every tile)
#pragma unroll
      for (int i = 0; i < RB; ++i) a3[i] = *reinterpret_cast<const f16x8*>(sA + a_row + i * 256 + c_alo);
#pragma unroll
      for (int j = 0; j < CB; ++j) b3[j] = b1[j] * kDown;
    }
    {
      const float* sA = ring_step(it + 1);
      if (g >= 2) {
#pragma unroll
        for (int i = 0; i < RB; ++i) a3[i] = *reinterpret_cast<const f16x8*>(sA + a_row + i * 256 + c_alo);
#pragma unroll
        for (int j = 0; j < CB; ++j) b3[j] = *reinterpret_cast<const f16x8*>(sA + w_row + j * 256 + c_whi) * kDown;
      }
      if constexpr (!DEFER) {
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
          for (int j = 0; j < CB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a3[i], b3[j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);  // (the I1 operands are read after the I3 instructions have been issued, not before)
      }
      f16x8 a1[RB], b1[CB];
      read_i1(sA, a1, b1);
      issue_i1(a1, b1);
    }

    kt += 2;
    if (kt == nk) {
      if constexpr (DEFER) {  // the last pair's I3; then zero operands for the next tile's first pair
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
          for (int j = 0; j < CB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a3[i], b3[j], acc[i][j], 0, 0, 0);
        clear_i3();
      }
      // ---- epilogue of the finished tile, straight from the accumulators: block (i, j) register r is row 16 i + 4 g + r, column
      // 32 (j >> 1) + 2 r16 + (j & 1) of the wave tile - a lane owns the column PAIR (c, c + 1), c = 32 q + 2 r16, in blocks 2q, 2q + 1 ----
      kt = 0;
      bool bad = false;
      const unsigned rows_left = (unsigned)min(p.M - m_base, BM);
      // the descriptor ends after row M - 1 and the block's position is added to the LANE offset (the range check that drops the
      // rows beyond M covers the lane offset only)
      const __amdgpu_buffer_rsrc_t rsrc_c = __builtin_amdgcn_make_buffer_rsrc(p.C + (size_t)m_base * p.ldc, 0, rows_left * (unsigned)p.ldc * 4u, 0x00020000);
      const __amdgpu_buffer_rsrc_t rsrc_r =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(RES ? p.R + (size_t)m_base * p.ldc : p.C), 0, rows_left * (unsigned)p.ldc * 4u, 0x00020000);
      constexpr int QB = CB / 2;  // column-pair blocks of 32 columns
      if constexpr (OPACK) {
        // pre-split result [row][N/16][16 hi | 16 lo*2^11] f16: the pair (c, c + 1) is one dword of hi halves and one of lo halves,
        // 32 bytes apart, inside the 64 bytes of the 16-column group c >> 4
        unsigned poff[4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
          poff[r] = ((unsigned)(wm * WM + 4 * g + r) * p.ldc + (unsigned)(n_base + wn * WN)) * 4u + (unsigned)((r16 >> 3) * 64 + (r16 & 7) * 4);
#pragma unroll
        for (int q = 0; q < QB; ++q) {
          if (n_base + wn * WN + q * 32 >= p.N) continue;  // (N % 32 == 0: a 32-column block is all in or all out)
#pragma unroll
          for (int i = 0; i < RB; ++i) {
            const unsigned boff = (unsigned)(i * 16) * p.ldc * 4u + (unsigned)(q * 128);  // wave-uniform
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              f32x2 v = {acc[i][2 * q][r] * w_down[2 * q], acc[i][2 * q + 1][r] * w_down[2 * q + 1]};
              if (ACT == 1) v = gelu_erf2(v);
              const float x0 = pinned(v.x), x1 = pinned(v.y);
              const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
              bad = bad || nonfinite((float)h0) || nonfinite((float)h1);  // also a finite x beyond f16's 65504
              const _Float16 l0 = (_Float16)((x0 - (float)h0) * 2048.0f), l1 = (_Float16)((x1 - (float)h1) * 2048.0f);
              const unsigned hw = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
              const unsigned lw = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
              __builtin_amdgcn_raw_buffer_store_b32(hw, rsrc_c, poff[r] + boff, 0, 2);
              __builtin_amdgcn_raw_buffer_store_b32(lw, rsrc_c, poff[r] + boff + 32u, 0, 2);
            }
          }
        }
        report_nonfinite(p.oflow, bad);
        li += gx;
        if (li < chunk_len) tile_coords(chunk_start + li, m_base, n_base);
        continue;
      }
      // fp32 result (+ residual): 8-byte accesses, 16 lanes = 128 contiguous bytes of a row, 4 rows per instruction; column pairs past
      // N are dropped by a per-lane predicate (N is even here: K % 32 == 0 shapes of the path; a lone last column takes the dword path)
      typedef unsigned u32x2m __attribute__((ext_vector_type(2)));
      unsigned voff[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) voff[r] = ((unsigned)(wm * WM + 4 * g + r) * p.ldc + (unsigned)(n_base + wn * WN + 2 * r16)) * 4u;
      // units of (block row, half of the column-pair blocks): the residual of unit u + 1 is requested before unit u is stored
      constexpr int HQ = QB > 1 ? QB / 2 : 1, UPR = QB / HQ, NU = RB * UPR;
      f32x2 rvb[2][HQ][4];
      auto unit_off = [&](int u, int qq) __attribute__((always_inline)) {  // wave-uniform byte offset of pair block (u / UPR, (u % UPR) HQ + qq)
        return (unsigned)((u / UPR) * 16) * p.ldc * 4u + (unsigned)((((u % UPR) * HQ) + qq) * 128);
      };
      auto res_load = [&](int u, f32x2 (&dst)[HQ][4]) __attribute__((always_inline)) {
#pragma unroll
        for (int qq = 0; qq < HQ; ++qq)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            dst[qq][r] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc_r, voff[r] + unit_off(u, qq), 0, 0));
      };
      if (RES) res_load(0, rvb[0]);
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        if (RES && u + 1 < NU) res_load(u + 1, rvb[(u + 1) & 1]);
        const int i = u / UPR;
#pragma unroll
        for (int qq = 0; qq < HQ; ++qq) {
          const int q = (u % UPR) * HQ + qq;
          const int col = n_base + wn * WN + q * 32 + 2 * r16;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            f32x2 v = {acc[i][2 * q][r] * w_down[2 * q], acc[i][2 * q + 1][r] * w_down[2 * q + 1]};
            if (ACT == 1) v = gelu_erf2(v);
            if (RES) v += rvb[u & 1][qq][r];
            const float vx = v.x, vy = v.y;
            if (col + 1 < p.N) {
              bad = bad || nonfinite(vx) || nonfinite(vy);
              __builtin_amdgcn_raw_buffer_store_b64(u32x2m{__builtin_bit_cast(unsigned, vx), __builtin_bit_cast(unsigned, vy)}, rsrc_c, voff[r] + unit_off(u, qq), 0, 2);
            } else if (col < p.N) {
              bad = bad || nonfinite(vx);
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vx), rsrc_c, voff[r] + unit_off(u, qq), 0, 2);
            }
          }
        }
      }
      report_nonfinite(p.oflow, bad);
      li += gx;
      if (li < chunk_len) tile_coords(chunk_start + li, m_base, n_base);
    }
  }
  if (probe) {
    atomicAdd(&p.clk[0], (unsigned long long)((long long)__builtin_readcyclecounter() - pc0));
    atomicAdd(&p.clk[1], (unsigned long long)((long long)wall_clock64() - pw0));
  }
}

template <int TM, int TN, int ACT, bool RES, bool OPACK>
int launch_one(const SplitParams& p, int grid, hipStream_t stream) {
  using Cfg = M16Cfg<TM, TN>;
  static std::atomic<unsigned long long> done{0};
  PMCE_TRY(pmce_opt_in_lds(reinterpret_cast<const void*>(&gemm_split_m16_kernel<TM, TN, ACT, RES, OPACK>), Cfg::LDS_BYTES, done, "gemm_split_m16"));
  hipLaunchKernelGGL((gemm_split_m16_kernel<TM, TN, ACT, RES, OPACK>), dim3(grid), dim3(256), Cfg::LDS_BYTES, stream, p);
  return PMCE_OK;
}
template <int TM, int TN>
int launch_cfg(SplitParams& p, int act, bool opack, hipStream_t stream) {
  using Cfg = M16Cfg<TM, TN>;
  p.ntm = (p.M + Cfg::BM - 1) / Cfg::BM;
  p.ntn = (p.N + Cfg::BN - 1) / Cfg::BN;
  const int per_cu = (160 * 1024) / Cfg::LDS_BYTES >= 3 ? 3 : 2;
  int g = p.ntm * p.ntn;
  if (g > 256 * per_cu) g = 256 * per_cu;
  g = (g + 7) & ~7;
  const bool res = p.R != nullptr;
  if (opack) return launch_one<TM, TN, 1, false, true>(p, g, stream);
  if (act == 1) return res ? launch_one<TM, TN, 1, true, false>(p, g, stream) : launch_one<TM, TN, 1, false, false>(p, g, stream);
  return res ? launch_one<TM, TN, 0, true, false>(p, g, stream) : launch_one<TM, TN, 0, false, false>(p, g, stream);
}

}  // namespace

// pre-split A, no row map, K a multiple of 32 (an even number of k-tiles), and - as everywhere - ldc bounded for the 32-bit offsets
bool pmce_gemm_split_m16_wants(int K, int a_packed, int c_div) { return a_packed && c_div == 0 && K % 32 == 0; }
int pmce_gemm_split_m16_launch(SplitParams& p, int act, int c_packed, int tile, hipStream_t stream) {
  switch (tile) {
    case 0: return launch_cfg<2, 4>(p, act, c_packed != 0, stream);
    case 1: return launch_cfg<2, 2>(p, act, c_packed != 0, stream);
    default: return launch_cfg<1, 2>(p, act, c_packed != 0, stream);
  }
}
