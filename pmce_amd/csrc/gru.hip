// Bidirectional 2-layer GRU (reference CoevoDecoder.py:216-221,228: nn.GRU(2048,1024,bidirectional, num_layers=2),
// seq-first, h0 = 0).  The input projections gi = W_ih x + b_ih are gemm_f32.hip launches; one time step is
//   gh = W_hh h + b_hh ; r = sigmoid(gi_r + gh_r) ; z = sigmoid(gi_z + gh_z) ; n = tanh(gi_n + r * gh_n) ;
//   h' = (1 - z) * n + z * h                                   (PyTorch gate order r, z, n).
#include <stdlib.h>

#include "common.hpp"

// ------------------------------------------------------------------------------------------------------
// Fused GRU time step: gh = h_prev W_hh^T (fp32 MFMA) + gate update, one launch for up to two directions.
// Block = 64 batch rows x 32 hidden units x {r,z,n}; 4 waves = 2 row blocks x 2 K-halves (K = 1024 split in two so
// that B = 256 gives 4 x 32 x 2 = 256 blocks, one per CU, all four SIMDs busy); the K-halves meet in LDS and
// the gate math runs on the accumulator layout (col = unit, row = batch), so gh never touches HBM.
//
// With ONE wave per SIMD nothing hides a workgroup barrier, so the K loop has none: every wave stages its own
// operand slices (32 rows of h, 96 rows of W_hh, 32 k per stage) into a private two-stage LDS ring by LDS-DMA
// (buffer_load_dwordx4 ... lds: no VGPR round trip, descriptor + fixed per-lane offsets + the k offset in soffset, so the
// loop issues no vector ALU instruction at all) and only waits on its own vmcnt.  Rows are unpadded 128-byte lines with
// XOR-swizzled 16-byte chunks (conflict-free ds_read_b128), the k order inside each 8 is permuted as in gemm_f32.hip.
// The two row-block waves of a K-half fetch the same W_hh lines; the second hits in the CU's L1.
// ------------------------------------------------------------------------------------------------------
struct GruStepArgs {
  const float* gi[2];     // + row*gi_rs + gate*H + unit   (b_ih already added by the input-projection GEMM)
  const float* whh[2];    // [3H][H]
  const float* bhh[2];    // [3H]
  const float* hprev[2];  // + row*h_rs + unit ; null -> h = 0 (first step)
  float* hout[2];         // + row*h_rs + unit
  long long gi_rs, h_rs;
  int B, H;
  const float* wscale;    // SPLIT form: [2 * 3H] 2^-s per row of the packed W_hh (direction-major, as packed: pmce_gemm_pack_split_f16)
  int wblk;               // SPLIT form: the packed W_hh is in the BLOCKED layout [3H / 64][H / 16][64 rows][16 hi | 16 lo] (pmce_gemm_pack_split_f16_blk): a
                          // DMA instruction's 16 rows of a k-tile are 1 KB contiguous (8 full cache lines) instead of 16 half lines one weight row apart -
                          // with every CU streaming, the L2 -> LDS path delivers 58 instead of 30 B/clk per CU (scripts/microbench/dma_patterns.hip)
};

typedef _Float16 gru_f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void gru_dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, int soff, unsigned lds_wave_base) {
  unsigned keep;  // M0 is compiler-reserved: saved and restored inside the statement
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(rsrc), "s"(lds_wave_base), "s"(soff)
      : "memory");
}

// fp32 pipe (the split mode's step is gru_step_v2_kernel below).  NQ = 4 K-slices per workgroup: 8 waves (two per SIMD: one wave's
// LDS / DMA waits hide under the other's MFMAs) staging 16 k per stage; the rings fill 128 KB of LDS.
__global__ __launch_bounds__(512) void gru_step_kernel(GruStepArgs a) {
  constexpr int NQ = 4;
  constexpr int KS = 64 / NQ;              // k per stage
  constexpr int ROWB = KS * 4;             // bytes per staged row
  constexpr int LPR = KS / 4;              // lanes (16-byte chunks) per row
  constexpr int RPI = 64 / LPR;            // rows per DMA instruction
  constexpr int NA = 32 / RPI, NW = 96 / RPI;
  constexpr int RW = 16 / NQ;              // accumulator rows each wave finishes
  constexpr int STAGE = (32 + 96) * ROWB;  // bytes per stage per wave: 32 h rows + 96 W rows
  // conflict-free ds_read_b128 on unpadded rows: chunk c of row r sits at c ^ swz(r)
  auto swz_of = [](int r) { return KS == 32 ? (r >> 1) & 7 : (r >> 2) & 3; };
  __shared__ __attribute__((aligned(16))) float smem[2 * NQ * 2 * STAGE / 4];  // 2*NQ waves x 2 stages = 128 KB
  const int d = blockIdx.z;
  const int u0 = blockIdx.x * 32, m0 = blockIdx.y * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n0 = lane & 31, hb = lane >> 5;
  const int rb = wave & 1, kq = wave >> 1;
  const int H = a.H, Kq = H / NQ;
  const float* __restrict__ hp = a.hprev[d];
  const float* __restrict__ W = a.whh[d];

  // This wave finishes RW of the 16 accumulator rows of its (row block, unit) tile: rows r = RW*kq .. RW*kq+RW-1.
  // Their gate inputs are fetched NOW so that they are in registers when the K loop ends.
  const int u = u0 + n0;
  const float* __restrict__ gi = a.gi[d];
  float gir[RW], giz[RW], gin[RW], hpv[RW];
#pragma unroll
  for (int q = 0; q < RW; ++q) {
    const int r = RW * kq + q;
    const int m = min(m0 + rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hb, a.B - 1);
    const float* g = gi + (long long)m * a.gi_rs + u;
    gir[q] = g[0];
    giz[q] = g[H];
    gin[q] = g[2 * H];
    hpv[q] = hp ? hp[(long long)m * a.h_rs + u] : 0.f;
  }

  f32x16 acc[3];
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;

  if (hp) {
    // DMA instruction i moves rows RPI*i .. of the stage (rows 0-31: h, rows 32-127: W gate-major): lane L lands at
    // row RPI*i + L/LPR, PHYSICAL chunk L%LPR, and fetches logical chunk (L%LPR) ^ swz(row).
    const __amdgpu_buffer_rsrc_t rsrc_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(hp), 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, 0xffffffff, 0x00020000);
    unsigned hoff[NA], woff[NW];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int row = RPI * i + lane / LPR;
      const int c = (lane % LPR) ^ swz_of(row);
      const int m = min(m0 + rb * 32 + row, a.B - 1);
      hoff[i] = (unsigned)(((long long)m * a.h_rs + 4 * c) * 4);
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int row = RPI * i + lane / LPR;  // gate = row >> 5, unit = row & 31
      const int c = (lane % LPR) ^ swz_of(row);
      woff[i] = (unsigned)(((long long)((row >> 5) * H + u0 + (row & 31)) * H + 4 * c) * 4);
    }
    const unsigned lds_wave =
        __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)smem) + wave * (2 * STAGE);
    auto dma_stage = [&](int kt) {
      const int soff = (kq * Kq + kt * KS) * 4;
      const unsigned dst = lds_wave + (kt & 1) * STAGE;
#pragma unroll
      for (int i = 0; i < NA; ++i) gru_dma16(rsrc_h, hoff[i], soff, dst + i * 1024);
#pragma unroll
      for (int i = 0; i < NW; ++i) gru_dma16(rsrc_w, woff[i], soff, dst + 32 * ROWB + i * 1024);
    };
    const int nk = Kq / KS;
    const float* my = smem + wave * (2 * STAGE / 4);
    const int swz = swz_of(n0);  // the same for rows n0, 32+n0, 64+n0, 96+n0
    dma_stage(0);
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) {
        dma_stage(kt + 1);
        // all but the NA+NW just issued: stage kt has landed
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      const float* As = my + (kt & 1) * (STAGE / 4) + n0 * KS;
      const float* Bs = my + (kt & 1) * (STAGE / 4) + 32 * KS + n0 * KS;
#pragma unroll
      for (int g8 = 0; g8 < KS / 8; ++g8) {
        const int co = 4 * ((2 * g8 + hb) ^ swz);
        const f32x4 av = *reinterpret_cast<const f32x4*>(As + co);
        f32x4 bv[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) bv[g] = *reinterpret_cast<const f32x4*>(Bs + g * 32 * KS + co);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int g = 0; g < 3; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[g][s], acc[g], 0, 0, 0);
      }
    }
    __syncthreads();  // every wave is done with its private ring: the reduction below reuses the memory
    // meet the K-slices: every wave publishes the rows its partners finish, then sums the partners' copies of its own
    float* red = smem;  // [dst kq][src slot][rb][gate][q][lane]
#pragma unroll
    for (int dq = 0; dq < NQ; ++dq) {
      if (dq == kq) continue;
      const int slot = kq < dq ? kq : kq - 1;  // index of this wave among dq's NQ-1 partners
#pragma unroll
      for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int q = 0; q < RW; ++q) {
          float v = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) v = (r == RW * dq + q) ? acc[g][r] : v;  // static register index, selected at run time
          red[((((dq * (NQ - 1) + slot) * 2 + rb) * 3 + g) * RW + q) * 64 + lane] = v;
        }
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int q = 0; q < RW; ++q) {
        float sum = 0.f;
#pragma unroll
        for (int sl = 0; sl < NQ - 1; ++sl) sum += red[((((kq * (NQ - 1) + sl) * 2 + rb) * 3 + g) * RW + q) * 64 + lane];
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (r == RW * kq + q) acc[g][r] += sum;
      }
  }
  // gate update on the D layout: unit = u0 + (lane & 31), batch row = m0 + rb*32 + (r&3) + 8*(r>>2) + 4*hb
  const float* __restrict__ bh = a.bhh[d];
  const float bhr = bh[u], bhz = bh[H + u], bhn = bh[2 * H + u];
  float* __restrict__ ho = a.hout[d];
#pragma unroll
  for (int q = 0; q < RW; ++q) {
    float ar = 0.f, az = 0.f, an = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const bool sel = (r == RW * kq + q);
      ar = sel ? acc[0][r] : ar;
      az = sel ? acc[1][r] : az;
      an = sel ? acc[2][r] : an;
    }
    const int r = RW * kq + q;
    const int m = m0 + rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hb;
    if (m < a.B) {
      const float rr = sigmoidf_acc(gir[q] + (ar + bhr));
      const float zz = sigmoidf_acc(giz[q] + (az + bhz));
      const float nn = tanhf(gin[q] + rr * (an + bhn));
      ho[(long long)m * a.h_rs + u] = (1.0f - zz) * nn + zz * hpv[q];
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// Round 4: the split-f16 step with the operand traffic cut by a third and a deeper ring.  Same tile (64 batch rows x 32 units x
// {r, z, n}), same 8 waves = 2 row blocks x 4 K-quarters, same arithmetic and summation order as gru_step_kernel<4, true> - results
// are bit-identical - but the two row-block waves of a K-quarter no longer stage PRIVATE copies of the quarter's 96 W_hh rows (each
// fetched them: 1.04 MB per workgroup and step for 0.65 MB of operands, and with two 8 KB stages per wave the 16 k-tiles were 16
// exposed round trips - the step's 21-24 us were the fill, `scripts/microbench/gru_ablate.py`).  Here a K-quarter owns ONE ring of
// three 10 KB stages [64 h rows | 96 W rows] x 64 B; each of its two waves DMAs its own 32 h rows and HALF of the W rows (5 instead
// of 8 instructions per k-tile), two k-tiles are in flight behind the one being multiplied, and one workgroup barrier per k-tile
// makes the partner's half visible (8 waves, two per SIMD: the barrier is cheap next to a round trip).  120 KB of LDS.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void gru_step_v2_kernel(GruStepArgs a) {
  constexpr int NQ = 4, NS = 3;
  constexpr int RW = 16 / NQ;                  // accumulator rows each wave finishes
  constexpr int STAGE = (64 + 96) * 64;        // bytes per stage: 160 rows of 64 B
  __shared__ __attribute__((aligned(16))) float smem[NQ * NS * STAGE / 4];   // 120 KB
  const int d = blockIdx.z;
  const int u0 = blockIdx.x * 32, m0 = blockIdx.y * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n0 = lane & 31, hb = lane >> 5;
  const int rb = wave & 1, kq = wave >> 1;
  const int H = a.H, Kq = H / NQ;
  const float* __restrict__ hp = a.hprev[d];
  const float* __restrict__ W = a.whh[d];

  const int u = u0 + n0;
  const float* __restrict__ gi = a.gi[d];
  float gir[RW], giz[RW], gin[RW], hpv[RW];
#pragma unroll
  for (int q = 0; q < RW; ++q) {
    const int r = RW * kq + q;
    const int m = min(m0 + rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hb, a.B - 1);
    const float* g = gi + (long long)m * a.gi_rs + u;
    gir[q] = g[0];
    giz[q] = g[H];
    gin[q] = g[2 * H];
    hpv[q] = hp ? hp[(long long)m * a.h_rs + u] : 0.f;
  }

  f32x16 acc[3];
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;

  if (hp) {
    const __amdgpu_buffer_rsrc_t rsrc_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(hp), 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, 0xffffffff, 0x00020000);
    // DMA instruction = 16 rows x 64 B (4 lanes per row).  This wave moves stage rows [32 rb, 32 rb + 32) (its h rows) and
    // [64 + 48 rb, 64 + 48 rb + 48) (its half of the W rows; W row r' = gate * 32 + unit).  Lane L lands at row + L / 4, PHYSICAL chunk
    // L % 4, and fetches logical chunk (L % 4) ^ ((row >> 2) & 3) (conflict-free ds_read_b128 on unpadded rows).
    unsigned hoff[2], woff[3];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = 32 * rb + 16 * i + (lane >> 2);
      const int c = (lane & 3) ^ ((row >> 2) & 3);
      const int m = min(m0 + row, a.B - 1);
      hoff[i] = (unsigned)(((long long)m * a.h_rs + 4 * c) * 4);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int r2 = 48 * rb + 16 * i + (lane >> 2);  // gate = r2 >> 5, unit = r2 & 31
      const int c = (lane & 3) ^ ((r2 >> 2) & 3);
      const int wr = (r2 >> 5) * H + u0 + (r2 & 31);  // row of this direction's [3H][H]
      woff[i] = a.wblk ? (unsigned)(((long long)(wr >> 6) * (H / 16) * 1024 + (wr & 63) * 16 + 4 * c) * 4) : (unsigned)(((long long)wr * H + 4 * c) * 4);
    }
    const int wks = a.wblk ? 4096 : 64;  // bytes from one k-tile of a W row (block) to the next
    const unsigned lds_q =
        __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)smem) + kq * (NS * STAGE);
    auto dma_stage = [&](int kt) {
      const int soff = (kq * Kq + kt * 16) * 4, soff_w = (kq * (Kq / 16) + kt) * wks;
      const unsigned dst = lds_q + (kt % NS) * STAGE;
#pragma unroll
      for (int i = 0; i < 2; ++i) gru_dma16(rsrc_h, hoff[i], soff, dst + (32 * rb + 16 * i) * 64);
#pragma unroll
      for (int i = 0; i < 3; ++i) gru_dma16(rsrc_w, woff[i], soff_w, dst + (64 + 48 * rb + 16 * i) * 64);
    };
    const int nk = Kq / 16;
    const float* ring = smem + kq * (NS * STAGE / 4);
    const int swz = (n0 >> 2) & 3;  // the same for stage rows n0, 32 + n0, 64 + n0, 96 + n0, 128 + n0
    dma_stage(0);
    dma_stage(1);
    for (int kt = 0; kt < nk; ++kt) {
      // k-tile kt has landed when at most the 5 DMAs of tile kt + 1 are still in flight (in-order completion)
      if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();  // the partner's half of tile kt is in LDS; every wave is done reading tile kt - 1
      if (kt + 2 < nk) dma_stage(kt + 2);  // into the stage tile kt - 1 occupied
      const float* st = ring + (kt % NS) * (STAGE / 4);
      const float* As = st + (32 * rb + n0) * 16;
      const float* Bs = st + (64 + n0) * 16;
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(As + 4 * ((2 * hb) ^ swz));      // k = 8 hb + [0, 4)
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(As + 4 * ((2 * hb + 1) ^ swz));  // k = 8 hb + [4, 8)
      const float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
      gru_f16x8 ahi, alo;
#pragma unroll
      for (int e = 0; e < 8; ++e) ahi[e] = (_Float16)xv[e];
#pragma unroll
      for (int e = 0; e < 8; ++e) alo[e] = (_Float16)((xv[e] - (float)ahi[e]) * 2048.0f);
      gru_f16x8 whi[3], wlo[3], wh2[3];
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        whi[g] = *reinterpret_cast<const gru_f16x8*>(Bs + g * 32 * 16 + 4 * (hb ^ swz));        // hi plane, k = 8 hb + [0, 8)
        wlo[g] = *reinterpret_cast<const gru_f16x8*>(Bs + g * 32 * 16 + 4 * ((2 + hb) ^ swz));  // lo plane
        wh2[g] = whi[g] * (_Float16)0.00048828125f;
      }
#pragma unroll
      for (int g = 0; g < 3; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, whi[g], acc[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 3; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, wlo[g], acc[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 3; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, wh2[g], acc[g], 0, 0, 0);
    }
    __syncthreads();  // every wave is done with the rings: the reduction below reuses the memory
    // meet the K-slices: every wave publishes the rows its partners finish, then sums the partners' copies of its own
    float* red = smem;  // [dst kq][src slot][rb][gate][q][lane]
#pragma unroll
    for (int dq = 0; dq < NQ; ++dq) {
      if (dq == kq) continue;
      const int slot = kq < dq ? kq : kq - 1;
#pragma unroll
      for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int q = 0; q < RW; ++q) {
          float v = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) v = (r == RW * dq + q) ? acc[g][r] : v;
          red[((((dq * (NQ - 1) + slot) * 2 + rb) * 3 + g) * RW + q) * 64 + lane] = v;
        }
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int q = 0; q < RW; ++q) {
        float sum = 0.f;
#pragma unroll
        for (int sl = 0; sl < NQ - 1; ++sl) sum += red[((((kq * (NQ - 1) + sl) * 2 + rb) * 3 + g) * RW + q) * 64 + lane];
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (r == RW * kq + q) acc[g][r] += sum;
      }
  }
  const float* __restrict__ bh = a.bhh[d];
  const float bhr = bh[u], bhz = bh[H + u], bhn = bh[2 * H + u];
  const float wd_r = a.wscale[d * 3 * H + u], wd_z = a.wscale[d * 3 * H + H + u], wd_n = a.wscale[d * 3 * H + 2 * H + u];
  float* __restrict__ ho = a.hout[d];
#pragma unroll
  for (int q = 0; q < RW; ++q) {
    float ar = 0.f, az = 0.f, an = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const bool sel = (r == RW * kq + q);
      ar = sel ? acc[0][r] : ar;
      az = sel ? acc[1][r] : az;
      an = sel ? acc[2][r] : an;
    }
    ar *= wd_r;
    az *= wd_z;
    an *= wd_n;
    const int r = RW * kq + q;
    const int m = m0 + rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hb;
    if (m < a.B) {
      const float rr = sigmoidf_acc(gir[q] + (ar + bhr));
      const float zz = sigmoidf_acc(giz[q] + (az + bhz));
      const float nn = tanhf(gin[q] + rr * (an + bhn));
      ho[(long long)m * a.h_rs + u] = (1.0f - zz) * nn + zz * hpv[q];
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// Round 5: the split-f16 step for SMALL batches (B <= 64: the decoder-only configuration at batch 64, streaming, single clips).  gru_step_v2
// gives a workgroup 32 units x 64 rows: B <= 64 is 64 workgroups on 256 CUs, each pulling 0.65 MB through ONE CU's L2 port behind 80 KB of
// requests in flight - 8 dependent round trips, 12-16 us per step whatever the batch.  Here a workgroup owns 8 units x {r, z, n} of one
// direction - the 24 W_hh rows are ONE 32-column matrix operand [r 8 | z 8 | n 8 | -] - for ALL batch rows: 256 workgroups, one per CU, each
// streaming 98 KB of W_hh + B x 4 KB of h.  Four waves = the four K-quarters of gru_step_v2, each with a PRIVATE ring (no barrier in the
// loop) of 8 / 5 stages of [32 NT h rows | 32 W rows] x 64 B, all but one stage in flight (112 / 96 KB per CU).  The arithmetic per output
// element is gru_step_v2's, operation for operation - the same k-tiles in the same order per K-quarter, the quarters met in the same order
// (the quarter that owns accumulator row r = the one v2's wave kq finishes) - so the result is bit-identical to it
// (test_gru_step_small_batch_equals_v2): a clip's numbers do not depend on the batch it rode in.
// ------------------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256) void gru_step_small_kernel(GruStepArgs a) {
  constexpr int NQ = 4;
  constexpr int NS = NT == 1 ? 8 : 5;
  constexpr int ROWS = 32 * NT + 32;           // rows of 64 B per stage: the h rows of NT batch tiles, then the 32 W rows
  constexpr int STAGE = ROWS * 64;             // bytes
  constexpr int NI = 2 * NT + 2;               // DMA instructions per stage
  __shared__ __attribute__((aligned(16))) float smem[NQ * NS * STAGE / 4];   // 128 KB / 120 KB
  const int d = blockIdx.z, u0 = blockIdx.x * 8;
  const int tid = threadIdx.x, lane = tid & 63, kq = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n0 = lane & 31, hb = lane >> 5;
  const int H = a.H, Kq = H / NQ;
  const float* __restrict__ hp = a.hprev[d];
  const float* __restrict__ W = a.whh[d];
  const float* __restrict__ gi = a.gi[d];

  // the gate phase's operands, requested before the loop: thread t finishes (row 32 nt + t / 8, unit u0 + t % 8)
  const int gm = tid >> 3, gu = u0 + (tid & 7);
  float gir[NT], giz[NT], gin[NT], hpv[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int m = min(32 * nt + gm, a.B - 1);
    const float* g = gi + (long long)m * a.gi_rs + gu;
    gir[nt] = g[0];
    giz[nt] = g[H];
    gin[nt] = g[2 * H];
    hpv[nt] = hp ? hp[(long long)m * a.h_rs + gu] : 0.f;
  }
  const float* __restrict__ bh = a.bhh[d];
  const float bhr = bh[gu], bhz = bh[H + gu], bhn = bh[2 * H + gu];
  const float wd_r = a.wscale[d * 3 * H + gu], wd_z = a.wscale[d * 3 * H + H + gu], wd_n = a.wscale[d * 3 * H + 2 * H + gu];

  float fin[NT][3];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) fin[nt][0] = fin[nt][1] = fin[nt][2] = 0.f;

  if (hp) {
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
    const __amdgpu_buffer_rsrc_t rsrc_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(hp), 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, 0xffffffff, 0x00020000);
    // DMA instruction = 16 rows x 64 B (4 lanes per row); lane L lands at row + L / 4, PHYSICAL chunk L % 4, and fetches logical chunk
    // (L % 4) ^ ((row >> 2) & 3) (conflict-free ds_read_b128 on unpadded rows), as in gru_step_v2.
    unsigned hoff[2 * NT], woff[2];
#pragma unroll
    for (int i = 0; i < 2 * NT; ++i) {
      const int row = 16 * i + (lane >> 2);
      const int c = (lane & 3) ^ ((row >> 2) & 3);
      const int m = min(row, a.B - 1);
      hoff[i] = (unsigned)(((long long)m * a.h_rs + 4 * c) * 4);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int col = 16 * i + (lane >> 2);          // operand column: gate = col >> 3 (3 = unused: a copy of n), unit = col & 7
      const int c = (lane & 3) ^ ((col >> 2) & 3);
      const int wr = min(col >> 3, 2) * H + u0 + (col & 7);
      woff[i] = a.wblk ? (unsigned)(((long long)(wr >> 6) * (H / 16) * 1024 + (wr & 63) * 16 + 4 * c) * 4) : (unsigned)(((long long)wr * H + 4 * c) * 4);
    }
    const int wks = a.wblk ? 4096 : 64;
    const unsigned lds_q =
        __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)smem) + kq * (NS * STAGE);
    auto dma_stage = [&](int kt) {
      const int soff = (kq * Kq + kt * 16) * 4, soff_w = (kq * (Kq / 16) + kt) * wks;
      const unsigned dst = lds_q + (kt % NS) * STAGE;
#pragma unroll
      for (int i = 0; i < 2 * NT; ++i) gru_dma16(rsrc_h, hoff[i], soff, dst + 16 * i * 64);
#pragma unroll
      for (int i = 0; i < 2; ++i) gru_dma16(rsrc_w, woff[i], soff_w, dst + (32 * NT + 16 * i) * 64);
    };
    const int nk = Kq / 16;
    const float* ring = smem + kq * (NS * STAGE / 4);
    const int swz = (n0 >> 2) & 3;
    for (int kt = 0; kt < NS - 1 && kt < nk; ++kt) dma_stage(kt);
    for (int kt = 0; kt < nk; ++kt) {
      // tile kt has landed when at most the NS - 2 tiles issued after it are in flight (in-order completion); the tail waits for all
      if (kt + NS - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * NI) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (kt + NS - 1 < nk) dma_stage(kt + NS - 1);  // into the stage tile kt - 1 occupied (its fragment reads were issued an iteration ago)
      const float* st = ring + (kt % NS) * (STAGE / 4);
      const float* Bs = st + (32 * NT + n0) * 16;
      const gru_f16x8 whi = *reinterpret_cast<const gru_f16x8*>(Bs + 4 * (hb ^ swz));        // hi plane, k = 8 hb + [0, 8)
      const gru_f16x8 wlo = *reinterpret_cast<const gru_f16x8*>(Bs + 4 * ((2 + hb) ^ swz));  // lo plane
      const gru_f16x8 wh2 = whi * (_Float16)0.00048828125f;
      gru_f16x8 ahi[NT], alo[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float* As = st + (32 * nt + n0) * 16;
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(As + 4 * ((2 * hb) ^ swz));
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(As + 4 * ((2 * hb + 1) ^ swz));
        const float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) ahi[nt][e] = (_Float16)xv[e];
#pragma unroll
        for (int e = 0; e < 8; ++e) alo[nt][e] = (_Float16)((xv[e] - (float)ahi[nt][e]) * 2048.0f);
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[nt], whi, acc[nt], 0, 0, 0);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[nt], wlo, acc[nt], 0, 0, 0);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[nt], wh2, acc[nt], 0, 0, 0);
    }
    __syncthreads();  // every wave is done with its ring: the exchange below reuses the memory
    float* red = smem;  // [kq][nt][r][lane]
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[((kq * NT + nt) * 16 + r) * 64 + lane] = acc[nt][r];
    __syncthreads();
    // row gm of a batch tile is accumulator register r = (gm & 3) + 4 ((gm >> 3) & 3) of the lanes hb = (gm >> 2) & 1; gru_step_v2's wave
    // kq = r / 4 finishes it as  own + ((0 + other_0) + other_1) + other_2  with the other quarters in ascending order.
    const int r = (gm & 3) + 4 * ((gm >> 3) & 3), own = (gm >> 3) & 3, lhb = 32 * ((gm >> 2) & 1);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        const int ln = 8 * g + (tid & 7) + lhb;
        float sum = 0.f, mine = 0.f;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const float v = red[((q * NT + nt) * 16 + r) * 64 + ln];
          if (q == own) mine = v;
          else sum += v;
        }
        fin[nt][g] = mine + sum;
      }
  }
  float* __restrict__ ho = a.hout[d];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int m = 32 * nt + gm;
    if (m < a.B) {
      const float ar = fin[nt][0] * wd_r, az = fin[nt][1] * wd_z, an = fin[nt][2] * wd_n;
      const float rr = sigmoidf_acc(gir[nt] + (ar + bhr));
      const float zz = sigmoidf_acc(giz[nt] + (az + bhz));
      const float nn = tanhf(gin[nt] + rr * (an + bhn));
      ho[(long long)m * a.h_rs + gu] = (1.0f - zz) * nn + zz * hpv[nt];
    }
  }
}

static int gru_step_any(const float* gi0, const float* gi1, const float* whh0, const float* whh1, const float* wscale,
                        const float* bhh0, const float* bhh1, const float* hp0, const float* hp1, float* ho0, float* ho1,
                        long long gi_rs, long long h_rs, int B, int H, int ndir, hipStream_t stream, int wblk = 0) {
  PMCE_REQUIRE(ndir == 1 || ndir == 2, "gru_step: ndir must be 1 or 2");
  PMCE_REQUIRE(gi0 && whh0 && bhh0 && ho0 && B > 0, "gru_step: null pointer");
  PMCE_REQUIRE(ndir == 1 || (gi1 && whh1 && bhh1 && ho1), "gru_step: second direction pointers missing");
  PMCE_REQUIRE(H > 0 && H % 256 == 0 && h_rs % 4 == 0, "gru_step: H must be a multiple of 256, h_rs of 4");
  GruStepArgs a;
  a.gi[0] = gi0; a.gi[1] = gi1; a.whh[0] = whh0; a.whh[1] = whh1; a.bhh[0] = bhh0; a.bhh[1] = bhh1;
  a.hprev[0] = hp0; a.hprev[1] = hp1; a.hout[0] = ho0; a.hout[1] = ho1;
  a.gi_rs = gi_rs; a.h_rs = h_rs; a.B = B; a.H = H; a.wscale = wscale; a.wblk = wblk;
  PMCE_REQUIRE((long long)B * h_rs * 4 < (1ll << 32) && 3ll * H * H * 4 < (1ll << 32), "gru_step: h or W_hh spans 4 GiB or more");
  if (wscale && B <= 32)
    hipLaunchKernelGGL(gru_step_small_kernel<1>, dim3(H / 8, 1, ndir), dim3(256), 0, stream, a);
  else if (wscale && B <= 64)
    hipLaunchKernelGGL(gru_step_small_kernel<2>, dim3(H / 8, 1, ndir), dim3(256), 0, stream, a);
  else if (wscale)
    hipLaunchKernelGGL(gru_step_v2_kernel, dim3(H / 32, (B + 63) / 64, ndir), dim3(512), 0, stream, a);
  else
    hipLaunchKernelGGL(gru_step_kernel, dim3(H / 32, (B + 63) / 64, ndir), dim3(512), 0, stream, a);
  return pmce_check_launch("gru_step");
}
extern "C" int pmce_gru_step_f32(const float* gi0, const float* gi1, const float* whh0, const float* whh1, const float* bhh0,
                                 const float* bhh1, const float* hp0, const float* hp1, float* ho0, float* ho1,
                                 long long gi_rs, long long h_rs, int B, int H, int ndir, hipStream_t stream) {
  return gru_step_any(gi0, gi1, whh0, whh1, nullptr, bhh0, bhh1, hp0, hp1, ho0, ho1, gi_rs, h_rs, B, H, ndir, stream);
}
// The same step with gh = h W_hh^T in the three-product f16 form: whh0 / whh1 are rows of ONE weight packed by
// pmce_gemm_pack_split_f16 (K = H), wscale its scale pair.
extern "C" int pmce_gru_step_split_f32(const float* gi0, const float* gi1, const float* whh0p, const float* whh1p,
                                       const float* wscale, const float* bhh0, const float* bhh1, const float* hp0,
                                       const float* hp1, float* ho0, float* ho1, long long gi_rs, long long h_rs, int B, int H,
                                       int ndir, hipStream_t stream) {
  PMCE_REQUIRE(wscale, "gru_step_split: null wscale");
  return gru_step_any(gi0, gi1, whh0p, whh1p, wscale, bhh0, bhh1, hp0, hp1, ho0, ho1, gi_rs, h_rs, B, H, ndir, stream);
}

// The same on a W_hh packed by pmce_gemm_pack_split_f16_blk (both directions' rows in ONE call, N = 6H, or one direction's 3H: a direction
// starts on a 64-row block either way): what the model runs - a DMA instruction's rows are contiguous.  Same numbers, bit for bit.
extern "C" int pmce_gru_step_split_blk_f32(const float* gi0, const float* gi1, const float* whh0b, const float* whh1b,
                                           const float* wscale, const float* bhh0, const float* bhh1, const float* hp0,
                                           const float* hp1, float* ho0, float* ho1, long long gi_rs, long long h_rs, int B, int H,
                                           int ndir, hipStream_t stream) {
  PMCE_REQUIRE(wscale, "gru_step_split_blk: null wscale");
  PMCE_REQUIRE(H % 64 == 0, "gru_step_split_blk: H must be a multiple of 64");
  return gru_step_any(gi0, gi1, whh0b, whh1b, wscale, bhh0, bhh1, hp0, hp1, ho0, ho1, gi_rs, h_rs, B, H, ndir, stream, 1);
}

// joints(m) = pose3d(mm) / 1000   (reference PMCE.py:18 — a true division, kept as one)
__global__ __launch_bounds__(256) void scale_div_kernel(const float* __restrict__ x, float* __restrict__ y, long long n,
                                                        float denom) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = x[i] / denom;
}

extern "C" int pmce_div_scalar_f32(const float* x, float* y, long long n, float denom, hipStream_t stream) {
  PMCE_REQUIRE(x && y && n > 0, "div_scalar: bad args");
  hipLaunchKernelGGL(scale_div_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, y, n, denom);
  return pmce_check_launch("div_scalar");
}
