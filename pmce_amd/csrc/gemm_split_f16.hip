// Three-product f16 "NT" GEMM for gfx950:  C[m,n] = epi( sum_k A[m,k] * W[n,k] + bias[n] )  (+ R[m,n]),  fp32 in memory.
//
// The pose lifter's Linear layers (reference PoseEstimation.py:13-29 via timm Attention/Mlp, and imgfeat_embed,
// PoseEstimation.py:80) are 25 of the path's 30 large products and 70 % of its time on the fp32 matrix pipe
// (v_mfma_f32_32x32x2_f32: 157 TFLOP/s).  The f16 pipe is 16x faster (v_mfma_f32_32x32x16_f16: 2.5 PFLOP/s) and
// accumulates in fp32, so an fp32 product costs three f16 products at fp32 accuracy:
//     a = ahi + alo * 2^-11,  w * 2^s = whi + wlo            (hi = rne16(x), lo = rne16(x - hi); 22 mantissa bits each)
//     a * w * 2^s  =  ahi whi + ahi wlo + alo (whi 2^-11)  +  O(2^-22)
// (the dropped lo*lo term and the planes' own rounding sit below the fp32 product's accumulation error: measured against an
// fp64 product the result is CLOSER than the fp32 pipe's, tests/test_gpu_ops.py).  2^s - a power of two per OUTPUT ROW of W
// (round 3; per tensor before), chosen by pmce_gemm_pack_split_f16 - lifts the row's max|w| to [2^14, 2^15) so that wlo and
// whi 2^-11 are normal f16 numbers for every weight within 2^-17 of the row's largest: an outlier row cannot starve the other
// rows' lo planes.  alo carries 2^11 so that it is normal wherever ahi is.  No operand relies on f16 sub-normals.
//
// Kernel: 256 threads = 2x2 waves, wave tile (32 TM) x (32 TN), block tile (64 TM) x (64 TN), k-tile 16.
//  * W is PACKED once: per row and 16-wide k-tile, 16 f16 hi then 16 f16 lo (64 bytes - the size of a 16-wide fp32 k-tile of
//    A, so both operands move as 64-byte rows).  A stays fp32 in memory (or arrives packed the same way, APACK) and is split in
//    registers between its ds_read and the matrix pipe.
//  * Tiles go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds), 16 rows x 64 B per wave instruction, into a ring of
//    NS = 3 or 4 stages; the 16-byte chunks of a row are XOR-swizzled with (row >> 2) & 3 on the DMA's source address and on
//    the ds_read address (conflict-free ds_read_b128).  A k-iteration is 24 (TN = 4) or 12 matrix instructions per wave -
//    0.3 us - far less than a loaded L2 / HBM round trip, hence NS - 1 k-tiles in flight and ONE barrier per iteration:
//        wait(k-tile it landed) ; barrier ; issue DMA of k-tile it+NS-1 into the stage read in iteration it-1 ; compute.
//    The stream of k-tiles runs across tile boundaries (the next tile's first k-tiles fly under this tile's last ones).
//    The tile's bias slice rides in with its first k-tile (one more DMA instruction of wave 0) and becomes the accumulators'
//    initial value: the k-loop has no compiler-counted vector-memory operation whose wait would drain the ring.
//  * PERSISTENT workgroups, two per CU, walk an XCD-local chunk of the tile order (as gemm_f32.hip); the epilogue (scale by
//    2^-s, GELU, residual, stores) runs straight from the accumulators while the CU's other workgroup keeps the pipe busy.
#include <atomic>

#include "gemm_split_common.hpp"

template <int TM, int TN>
struct SplitCfg {
  static constexpr int BM = 64 * TM, BN = 64 * TN;
  static constexpr int STAGE_FLOATS = (BM + BN) * 16;
#ifdef PMCE_SPLIT_NS3
  static constexpr int NS = 3;
#else
  static constexpr int NS = STAGE_FLOATS * 4 * 4 <= 64 * 1024 ? 4 : 3;
#endif
  static constexpr int LDS_BYTES = NS * STAGE_FLOATS * 4 + 4 * 1024;  // + two {bias, 2^-s} slice pairs (this tile's, the next one's)
  static constexpr int DPW = (BM + BN) / 64;  // DMA instructions per wave per k-tile (16 rows each)
};

// RS: A is packed AND row-scaled (raw inputs of any fp32 magnitude: imgfeat_embed, the GRU layer-0 input projection) - the
// accumulators start at zero and the epilogue computes acc * 2^-s(n) * 2^e(m) + bias[n].
// KSUB = 2 (round 4; the 64x128 tile on small grids): a stage holds TWO consecutive 16-wide k-tiles, each in the layout of a KSUB = 1
// stage, fetched together and multiplied behind ONE wait and ONE barrier.  A launch of a few hundred small tiles is bound by a
// workgroup's serial chain per k-tile (wait -> barrier -> DMA issue -> fragment reads -> 6 dependent matrix instructions per wave: 0.43 us
// whatever the grid, profiles/r04_c_*); two k-tiles per trip halve the trips.  Same arithmetic and k order: bit-identical results.
// LNEP (round 4; the N = 256 products of a C = 256 lifter block, proj and fc2): the tile is 64 x 256 - whole rows - and the epilogue is
// the LayerNorm chain that followed the product as a launch of its own (lifter.hip ln_chain: norm2 after proj; the shared post-norm
// and the next block's norm1 after fc2): the residual stream and the pre-split LN output leave the accumulators directly, the fp32
// row is never re-read.  Row statistics: two-pass (mean, then centred squares) like ln_chain; a row's 256 columns sit in the two
// column waves of its row half - 32 lanes x 4 blocks each - so a statistic is 3 in-lane adds, a 32-lane all-reduce (4 DPP rotations
// inside the rows of 16 + one swizzle across them; every lane ends with the same bits) and one exchange of 16 floats per lane half
// with the partner wave through LDS.
__device__ __forceinline__ float half32_allsum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));  // row_ror:8
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));  // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));  // row_ror:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));  // row_ror:1
  v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401f));                     // lane ^ 16
  return v;
}

template <int TM, int TN, int ACT, bool RES, bool APACK, bool OPACK = false, bool RS = false, int KSUB = 1, bool LNEP = false>
__global__ __launch_bounds__(256, 2) void gemm_split_kernel(SplitParams p) {
  static_assert(!RS || (APACK && !OPACK && !RES), "a row-scaled A is a packed A; no packed result, no residual");
  static_assert(!LNEP || (TM == 1 && TN == 4 && ACT == 0 && RES && !OPACK && !RS && KSUB == 1), "the LayerNorm epilogue: 64 x 256 tile, residual form");
  using Cfg = SplitCfg<TM, TN>;
  constexpr int BM = Cfg::BM, BN = Cfg::BN, WM = 32 * TM, WN = 32 * TN;
  constexpr int SUBF = Cfg::STAGE_FLOATS;                         // one 16-wide k-tile of the stage
  constexpr int NS = KSUB == 1 ? Cfg::NS : 3, SF = KSUB * SUBF, DPW = KSUB * Cfg::DPW;
  static_assert(KSUB == 1 || KSUB == 2, "one or two k-tiles per stage (four measured: no further gain)");
  constexpr int GA = BM / 16;  // 16-row groups of the A part of a stage (a multiple of 4: every wave's first GA/4 DMAs are A's)
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n0 = lane & 31, hb = lane >> 5;
  const int wm = wave & 1, wn = wave >> 1;

  // ---- persistent workgroups on an XCD-local chunk of the grouped tile order (gemm_f32.hip) ----
  const int nblk = p.ntm * p.ntn;
  const int xcd = blockIdx.x & 7, bx = blockIdx.x >> 3, gx = gridDim.x >> 3;
  const int cq = nblk >> 3, cr = nblk & 7;
  const int chunk_start = xcd < cr ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq;
  const int chunk_len = cq + (xcd < cr ? 1 : 0);
  if (bx >= chunk_len) return;
  // (8 until round 6, as gemm_f32.hip has it; 4 measures 0.2-0.9 % faster on the lifter's products at C = 512 and, through what they leave in the
  // caches, 2.5 % on the ln_chain launches between them - equal at C = 256; 16 slower, 2 / 1 mixed: profiles/r06_p_*, r06_q_*)
  constexpr int GROUP_M = 4;
  const int per_group = GROUP_M * p.ntn;
  auto tile_coords = [&](int bid, int& mb, int& nb) {
    const int group = bid / per_group;
    const int first_m = group * GROUP_M;
    const int gsz = min(p.ntm - first_m, GROUP_M);
    mb = (first_m + (bid % per_group) % gsz) * BM;
    nb = ((bid % per_group) / gsz) * BN;
  };
  // clock probe (bench.py: the shader clock the chip sustains under this kernel; MI355X runs it power-limited far below 2.4 GHz)
  const bool probe = p.clk != nullptr && tid == 0;
  const long long pc0 = probe ? (long long)__builtin_readcyclecounter() : 0, pw0 = probe ? (long long)wall_clock64() : 0;
  const int my_tiles = (chunk_len - bx + gx - 1) / gx;
  const int nk = p.K / (16 * KSUB);  // stages per tile
  const int total = my_tiles * nk;
  // (Round 2 started every CU's second workgroup half a tile late so that the two would not reach their epilogues together, -8...-18 % then;
  // with today's kernel they drift apart on their own and the delay only cost: +1.1 % at C = 512 without it,
  // profiles/r04_j_gemm_start_skew_ab.txt.  Removed in round 5.)

  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, 0xffffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.W), 0, 0xffffffff, 0x00020000);
  // bounded: lanes past bias[N-1] read zeros
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.bias ? p.N * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wscale), 0, p.N * 4, 0x00020000);
  // RS: the tile's slice of row scales (bounded: rows past M - 1 read zeros; they are never stored)
  const __amdgpu_buffer_rsrc_t rsrc_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.rscale), 0, RS ? p.M * 4 : 0, 0x00020000);

  // ---- DMA side.  Instruction q of a wave moves row group g = wave + 4 q of a stage: lane L -> row 16 g + (L >> 2), PHYSICAL
  // chunk L & 3, which holds logical chunk (L & 3) ^ ((row >> 2) & 3) = (L & 3) ^ ((L >> 4) & 3). ----
  const int drow = lane >> 2;
  const unsigned dchunk = (unsigned)(((lane & 3) ^ ((lane >> 4) & 3)) * 4);  // floats
  constexpr int DPS = Cfg::DPW;  // DMA instructions per wave per 16-wide k-tile
  unsigned doff[DPS];
  auto set_ptrs = [&](int mb, int nb) {
#pragma unroll
    for (int q = 0; q < DPS; ++q) {
      const int g = wave + 4 * q;
      if (q < GA / 4)   // (== g < GA: GA is a multiple of 4, so a wave's first GA / 4 instructions are A's whatever the wave)
        doff[q] = ((unsigned)min(mb + 16 * g + drow, p.M - 1) * p.lda + dchunk) * 4u;
      else {
        const unsigned r = (unsigned)min(nb + 16 * (g - GA) + drow, p.N - 1);
        doff[q] = p.wblk ? ((r >> 6) * (unsigned)(p.K / 16) * 1024u + (r & 63u) * 16u + dchunk) * 4u   // 64-row block, 64-byte rows
                         : (r * (unsigned)p.K + dchunk) * 4u;
      }
    }
  };
  const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)lds;
  const unsigned lds_wave = __builtin_amdgcn_readfirstlane(lds0 + wave * 1024);
  const int kstep_w = p.wblk ? 4096 : 64;  // bytes from one k-tile of a W row (block) to the next
  auto issue = [&](int kt, int stage) {
    const unsigned stage_base = lds_wave + stage * (SF * 4);   // ONE scalar per k-tile: the instruction's slot is added as an immediate
#pragma unroll
    for (int sub = 0; sub < KSUB; ++sub) {
      const int ko = (kt * KSUB + sub) * 64, kw = (kt * KSUB + sub) * kstep_w;  // bytes
#pragma unroll
      for (int q = 0; q < DPS; ++q) {
        const bool is_a = q < GA / 4;   // compile-time: no run-time choice of descriptor and offset per instruction
        sdma16o(is_a ? rsrc_a : rsrc_w, doff[q], is_a ? ko : kw, stage_base + sub * (SUBF * 4), q * 4096);
      }
    }
  };

  // issue-side cursor (runs NS-1 k-tiles ahead of the compute side, across tile boundaries)
  int i_li = bx, i_kt = 0, i_stage = 0, issued = 0, i_nb = 0, i_mb = 0, i_par = 0;
  tile_coords(chunk_start + i_li, i_mb, i_nb);
  set_ptrs(i_mb, i_nb);
  auto issue_next = [&]() {
    if (i_kt == 0) {  // the tile's bias and 2^-s slices: 64 lanes x 4 floats each
      if (p.bias && wave == 0) sdma16(rsrc_b, (unsigned)lane * 16u, i_nb * 4, lds0 + NS * SF * 4 + i_par * 2048);
      if (wave == 1) sdma16(rsrc_s, (unsigned)lane * 16u, i_nb * 4, lds0 + NS * SF * 4 + i_par * 2048 + 1024);
      if constexpr (RS)  // and its BM row scales (behind the two slice pairs)
        if (wave == 2) sdma16(rsrc_rs, (unsigned)lane * 16u, i_mb * 4, lds0 + NS * SF * 4 + 4096 + i_par * 1024);
      i_par ^= 1;
    }
    issue(i_kt, i_stage);
    ++issued;
    i_stage = i_stage + 1 == NS ? 0 : i_stage + 1;
    if (++i_kt == nk) {
      i_kt = 0;
      i_li += gx;
      if (i_li < chunk_len) {
        tile_coords(chunk_start + i_li, i_mb, i_nb);
        set_ptrs(i_mb, i_nb);
      }
    }
  };
#pragma unroll
  for (int q = 0; q < NS - 1; ++q)
    if (issued < total) issue_next();

  float w_down[TN];  // 2^-s of this lane's column in each of the wave's TN 32-column blocks
  float b_late[TN];  // RS: the bias of this lane's columns, added after the row scaling
  const int swz = (n0 >> 2) & 3;
  const int a_row = (wm * WM + n0) * 16, w_row = BM * 16 + (wn * WN + n0) * 16;  // floats inside a stage
  // chunk offsets (floats) of this lane's operand fragments
  const int ca0 = 4 * ((2 * hb) ^ swz), ca1 = 4 * ((2 * hb + 1) ^ swz);  // fp32 A: k = 8 hb + [0,4), + [4,8)
  const int ch = 4 * (hb ^ swz), cl = 4 * ((2 + hb) ^ swz);              // packed: hi / lo plane, k = 8 hb + [0,8)

  f32x16 acc[TM][TN];
  int li = bx, kt = 0, stage = 0, c_par = 0;
  int m_base, n_base;
  tile_coords(chunk_start + li, m_base, n_base);

  for (int it = 0; it < total; ++it) {
    // k-tile `it` has landed when at most (younger batches) x DPW of this wave's DMAs are still in flight (in-order
    // completion; anything else outstanding only makes the wait stricter)
    const int younger = issued - it - 1;
    if (NS == 4 && younger >= 2) wait_vm<2 * DPW>();
    else if (younger >= 1) wait_vm<DPW>();
    else wait_vm<0>();
    __syncthreads();  // every wave's part of k-tile `it` is in LDS; every wave is done reading the stage of k-tile it-1
    if (issued < total) issue_next();

    if (kt == 0) {  // the bias slice (scaled like its row of W) is the accumulators' initial value
      const float* sB = lds + NS * SF + c_par * 512 + wn * WN + n0;
      c_par ^= 1;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        w_down[j] = sB[256 + j * 32];
        if constexpr (RS) b_late[j] = p.bias ? sB[j * 32] : 0.f;
        const float bv = (p.bias && !RS) ? sB[j * 32] * pow2_recip(w_down[j]) : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = bv;
      }
    }
#pragma unroll
    for (int sub = 0; sub < KSUB; ++sub) {
    const float* sA = lds + stage * SF + sub * SUBF;
    f16x8 ahi[TM], alo[TM], whi[TN], wlo[TN], wh2[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if constexpr (APACK) {
        ahi[i] = *reinterpret_cast<const f16x8*>(sA + a_row + i * 512 + ch);
        alo[i] = *reinterpret_cast<const f16x8*>(sA + a_row + i * 512 + cl);
      } else {
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(sA + a_row + i * 512 + ca0);
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(sA + a_row + i * 512 + ca1);
        split8(x0, x1, ahi[i], alo[i]);
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      whi[j] = *reinterpret_cast<const f16x8*>(sA + w_row + j * 512 + ch);
      wlo[j] = *reinterpret_cast<const f16x8*>(sA + w_row + j * 512 + cl);
      wh2[j] = whi[j] * (_Float16)0.00048828125f;  // 2^-11: undoes the scale of alo
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[i], whi[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[i], wlo[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[i], wh2[j], acc[i][j], 0, 0, 0);
    }

    stage = stage + 1 == NS ? 0 : stage + 1;
    if (++kt == nk) {
      // ---- epilogue of the finished tile, straight from the accumulators ----
      kt = 0;
      bool bad = false;  // this lane produced a non-finite value (an operand beyond the f16 range, or fp32 overflow)
      if constexpr (LNEP) {
        // rows of this lane: m_base + 32 wm + 4 hb + (r & 3) + 8 (r >> 2); columns 128 wn + 32 j + n0 (n_base = 0, N = ldc = 256)
        float* sX = lds + NS * SF + 1024;  // exchange slots [pass][wm][wn][hb][16]
        const int rows_left = min(p.M - m_base, BM);
        const unsigned row_bytes = 256u * 4u;
        const __amdgpu_buffer_rsrc_t rsrc_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.R + (size_t)m_base * 256), 0,
                                                                                 (unsigned)rows_left * row_bytes, 0x00020000);
        unsigned lane_off = (unsigned)(wm * 32 + 4 * hb) * row_bytes + (unsigned)(wn * 128 + n0) * 4u;
        asm volatile("" : "+v"(lane_off));  // (keeps the 16 + 64 offsets derived from it out of the k-loop's registers: made here, per tile)
        auto voff = [&](int r) __attribute__((always_inline)) { return lane_off + (unsigned)((r & 3) + 8 * (r >> 2)) * row_bytes; };
        // x = product * 2^-s + R (the bias came in through the accumulators); rows past M read zeros and are never stored
        {  // (the residual of block j + 1 is requested before block j is used: two 16-register sets, as in the plain epilogue)
          float rv[2][16];
          auto res_load = [&](int j, float (&dst)[16]) __attribute__((always_inline)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_r, voff(r), j * 128, 0));
          };
          res_load(0, rv[0]);
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            if (j + 1 < TN) res_load(j + 1, rv[(j + 1) & 1]);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float v = acc[0][j][r] * w_down[j] + rv[j & 1][r];
              bad = bad || nonfinite(v);
              acc[0][j][r] = v;
            }
          }
        }
        int pass = 0;
        // sum over the row of f(element) for this lane's 16 rows: both column waves end with the same 16 totals
        auto row_totals = [&](float (&t)[16]) __attribute__((always_inline)) {
#pragma unroll
          for (int r = 0; r < 16; ++r) t[r] = half32_allsum(t[r]);
          float* slot = sX + (((pass * 2 + wm) * 2 + wn) * 2 + hb) * 16;
          if (n0 == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(slot + 4 * q) = f32x4{t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]};
          }
          __syncthreads();
          const float* other = sX + (((pass * 2 + wm) * 2 + (wn ^ 1)) * 2 + hb) * 16;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 o = *reinterpret_cast<const f32x4*>(other + 4 * q);
            t[4 * q] += o.x; t[4 * q + 1] += o.y; t[4 * q + 2] += o.z; t[4 * q + 3] += o.w;
          }
          ++pass;
        };
        // acc <- LN(acc; w, b, eps) over the 256 columns of each row
        auto layer_norm = [&](const float* __restrict__ w, const float* __restrict__ b, float eps) __attribute__((always_inline)) {
          float t[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) t[r] = (acc[0][0][r] + acc[0][1][r]) + (acc[0][2][r] + acc[0][3][r]);
          row_totals(t);
          float mean[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) mean[r] = t[r] * (1.0f / 256.0f);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float q2 = 0.f;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              const float d = acc[0][j][r] - mean[r];
              acc[0][j][r] = d;
              q2 = fmaf(d, d, q2);
            }
            t[r] = q2;
          }
          row_totals(t);
#pragma unroll
          for (int r = 0; r < 16; ++r) t[r] = 1.0f / sqrtf(t[r] * (1.0f / 256.0f) + eps);
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const float wv = w[wn * 128 + j * 32 + n0], bv = b[wn * 128 + j * 32 + n0];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][j][r] = fmaf(acc[0][j][r] * t[r], wv, bv);
          }
        };
        if (p.ln1_w) layer_norm(p.ln1_w, p.ln1_b, p.ln1_eps);
        if (p.out1) {
          const __amdgpu_buffer_rsrc_t rsrc_o = __builtin_amdgcn_make_buffer_rsrc(p.out1 + (size_t)m_base * 256, 0, (unsigned)rows_left * row_bytes, 0x00020000);
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float y = acc[0][j][r];  // (a bit_cast applied to the vector component itself reads component 0 every time)
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y), rsrc_o, voff(r), j * 128, 2);
            }
        }
        if (p.out2) {
          layer_norm(p.ln2_w, p.ln2_b, p.ln2_eps);
          // pre-split [row][K/16][16 hi | 16 lo*2^11] f16 in the bytes of the fp32 row, as the OPACK epilogue writes it
          const bool odd = lane & 1;
          const unsigned psel = split_perm_sel(lane);
          const int colf = (n0 >> 4) * 32 + (odd ? 16 + ((n0 - 1) & 15) : (n0 & 15));
          unsigned lane_pk = (unsigned)(wm * 32 + 4 * hb) * row_bytes + (unsigned)(wn * 128) * 4u + (unsigned)colf * 2u;
          asm volatile("" : "+v"(lane_pk));
          const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(p.out2 + (size_t)m_base * 256, 0, (unsigned)rows_left * row_bytes, 0x00020000);
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const unsigned outw = split_pack_exchange(pinned(acc[0][j][r]), psel, bad);
              const unsigned vo = lane_pk + (unsigned)((r & 3) + 8 * (r >> 2)) * row_bytes + (unsigned)(j * 32) * 4u;
              __builtin_amdgcn_raw_buffer_store_b32(outw, rsrc_x, vo, 0, 0);  // write-back: the next product's A operand (as ln_chain's out2)
            }
        }
        report_nonfinite(p.oflow, bad);
        li += gx;
        if (li < chunk_len) tile_coords(chunk_start + li, m_base, n_base);
        continue;
      }
      if constexpr (RS) {
        // Row-scaled A: C[m][n] = acc * 2^-s(n) * 2^e(m) + bias[n].  2^e of the tile's rows landed with the tile's first k-tile
        // (c_par has moved on to the next tile's parity since).  Row-major over the accumulators - one row scale at a time serves
        // the TN column blocks - so the epilogue holds no table of scales in registers.
        const float* sR = lds + NS * SF + 1024 + (c_par ^ 1) * 256 + wm * WM + 4 * hb;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int rr = (r & 3) + 8 * (r >> 2);
            const int mm = m_base + wm * WM + i * 32 + 4 * hb + rr;
            const float up = sR[i * 32 + rr];
            if (mm < p.M) {
              float* crow = p.c_div > 0 ? p.C + (long long)(mm % p.c_div) * p.c_lo + (long long)(mm / p.c_div) * p.c_hi
                                        : p.C + (size_t)mm * p.ldc;
#pragma unroll
              for (int j = 0; j < TN; ++j) {
                const int n = n_base + wn * WN + j * 32 + n0;
                const float v = fmaf(acc[i][j][r] * w_down[j], up, b_late[j]);
                bad = bad || nonfinite(v);
                if (n < p.N) __builtin_nontemporal_store(v, crow + n);
              }
            }
          }
        report_nonfinite(p.oflow, bad);
        li += gx;
        if (li < chunk_len) tile_coords(chunk_start + li, m_base, n_base);
        continue;
      }
      if constexpr (OPACK) {
        // The result is itself the A operand of the next product (fc1 -> fc2): written pre-split, [row][K/16][16 hi | 16 lo*2^11]
        // f16 in the bytes of the fp32 row.  A lane holds ONE column of 16 rows; adjacent lanes pair up (DPP quad_perm) so that
        // every lane still stores one dword per element: even lanes {hi(n), hi(n+1)}, odd lanes {lo(n-1), lo(n)}.
        const bool odd = lane & 1;
        const unsigned psel = split_perm_sel(lane);
        const int colf = (n0 >> 4) * 32 + (odd ? 16 + ((n0 - 1) & 15) : (n0 & 15));  // f16 index inside the 32-column group
        // Buffer-form stores: 16 lane offsets (row r of a 32x32 block + this lane's f16 pair) serve every block; the block's position
        // is added to the lane offset (NOT passed as the scalar offset: the range check that drops the rows beyond M covers the
        // lane offset only), and the descriptor ends after row M - 1.
        const unsigned rows_left = (unsigned)min(p.M - m_base, BM);
        const __amdgpu_buffer_rsrc_t rsrc_c = __builtin_amdgcn_make_buffer_rsrc(p.C + (size_t)m_base * p.ldc, 0, rows_left * (unsigned)p.ldc * 4u, 0x00020000);
        unsigned voff[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) voff[r] = (unsigned)(4 * hb + (r & 3) + 8 * (r >> 2)) * (unsigned)p.ldc * 4u + (unsigned)colf * 2u;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int cb = n_base + wn * WN + j * 32;
          if (cb >= p.N) continue;  // (N % 32 == 0: a 32-column block is all in or all out)
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const unsigned so = ((unsigned)(wm * WM + i * 32) * (unsigned)p.ldc + (unsigned)cb) * 4u;  // wave-uniform
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
              f32x2 v = {acc[i][j][r] * w_down[j], acc[i][j][r + 1] * w_down[j]};
              if (ACT == 1) v = gelu_erf2(v);
#pragma unroll
              for (int e = 0; e < 2; ++e) {
                const unsigned outw = split_pack_exchange(pinned(e ? v.y : v.x), psel, bad);
                __builtin_amdgcn_raw_buffer_store_b32(outw, rsrc_c, voff[r + e] + so, 0, 2);
              }
            }
          }
        }
        report_nonfinite(p.oflow, bad);
        li += gx;
        if (li < chunk_len) tile_coords(chunk_start + li, m_base, n_base);
        continue;
      }
      const bool full = (m_base + BM <= p.M) && (n_base + BN <= p.N) && p.c_div == 0;
      if (p.c_div > 0) {  // mapped rows (GRU layer-0 input projection: (b,t) rows -> time-major)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int n = n_base + wn * WN + j * 32 + n0;
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int mm = m_base + wm * WM + i * 32 + 4 * hb + (r & 3) + 8 * (r >> 2);
              if (n < p.N && mm < p.M) {
                float v = acc[i][j][r] * w_down[j];
                if (ACT == 1) v = gelu_erf(v);
                bad = bad || nonfinite(v);
                p.C[(long long)(mm % p.c_div) * p.c_lo + (long long)(mm / p.c_div) * p.c_hi + n] = v;
              }
            }
        }
      } else if (full) {
        // Full tile, buffer-form accesses: ONE set of 16 lane offsets (row r of a 32x32 block, this lane's column) serves every
        // block of the tile and both R and C - the block's position goes into the scalar offset - instead of a 64-bit address pair
        // per row and block.  The residual of block b + 1 is requested BEFORE block b is stored (two 16-register sets): left to
        // itself every block starts with 16 dependent loads whose latency nothing hides - 8 exposed round trips per tile.
        constexpr int NBLK = TM * TN;
        const __amdgpu_buffer_rsrc_t rsrc_c = __builtin_amdgcn_make_buffer_rsrc(p.C + (size_t)m_base * p.ldc, 0, 0xffffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsrc_r =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(RES ? p.R + (size_t)m_base * p.ldc : p.C), 0, 0xffffffff, 0x00020000);
        unsigned voff[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) voff[r] = ((unsigned)(4 * hb + (r & 3) + 8 * (r >> 2)) * p.ldc + (unsigned)n0) * 4u;
        auto blk_off = [&](int b) __attribute__((always_inline)) {  // wave-uniform
          return ((unsigned)(wm * WM + (b % TM) * 32) * p.ldc + (unsigned)(n_base + wn * WN + (b / TM) * 32)) * 4u;
        };
        float rv[2][16];
        auto res_load = [&](int b, float (&dst)[16]) __attribute__((always_inline)) {
          const unsigned so = blk_off(b);
#pragma unroll
          for (int r = 0; r < 16; ++r) dst[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_r, voff[r], so, 0));
        };
        if (RES) res_load(0, rv[0]);
#pragma unroll
        for (int b = 0; b < NBLK; ++b) {
          const int j = b / TM, i = b % TM;
          if (RES && b + 1 < NBLK) res_load(b + 1, rv[(b + 1) & 1]);
          const unsigned so = blk_off(b);
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            f32x2 v = {acc[i][j][r] * w_down[j], acc[i][j][r + 1] * w_down[j]};
            if (ACT == 1) v = gelu_erf2(v);
            if (RES) v += f32x2{rv[b & 1][r], rv[b & 1][r + 1]};
            // aux 2 = nt: streaming stores measure 5-9 % faster than write-back ones here
            const float vx = v.x, vy = v.y;  // (a bit_cast applied to the vector component itself reads component 0 both times)
            bad = bad || nonfinite(vx) || nonfinite(vy);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vx), rsrc_c, voff[r], so, 2);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vy), rsrc_c, voff[r + 1], so, 2);
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int n = n_base + wn * WN + j * 32 + n0;
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const int mrow = m_base + wm * WM + i * 32 + 4 * hb;
            float* __restrict__ Cp = p.C + (size_t)mrow * p.ldc + n;
            const float* __restrict__ Rp = RES ? p.R + (size_t)mrow * p.ldc + n : nullptr;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {  // the same arithmetic as the full-tile path (results do not depend on tile shape)
              f32x2 v = {acc[i][j][r] * w_down[j], acc[i][j][r + 1] * w_down[j]};
                if (ACT == 1) v = gelu_erf2(v);
#pragma unroll
              for (int e = 0; e < 2; ++e) {
                const int rr = ((r + e) & 3) + 8 * ((r + e) >> 2);
                if (n < p.N && mrow + rr < p.M) {
                  float o = e ? v.y : v.x;
                  if (RES) o += Rp[rr * p.ldc];
                  bad = bad || nonfinite(o);
                  Cp[rr * p.ldc] = o;
                }
              }
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

// ---- launch ---------------------------------------------------------------------------------------------------------------
static std::atomic<int> g_split_tile{-1};  // pmce_gemm_split_set_tuning (tests sweep the tile shapes through it); -1 = automatic
extern "C" int pmce_gemm_split_set_tuning(int tile) {
  g_split_tile.store(tile, std::memory_order_relaxed);
  return PMCE_OK;
}

template <int TM, int TN, int ACT, bool RES, bool APACK, bool OPACK = false, bool RS = false, int KSUB = 1, bool LNEP = false>
static int launch_one(const SplitParams& p, int grid, hipStream_t stream) {
  using Cfg = SplitCfg<TM, TN>;
  static std::atomic<unsigned long long> done{0};
  // + two slices of BM row scales (RS) / the LayerNorm epilogue's exchange slots (LNEP)
  constexpr int LDS = (KSUB == 1 ? Cfg::LDS_BYTES : 3 * KSUB * Cfg::STAGE_FLOATS * 4 + 4096) + (RS || LNEP ? 2048 : 0);
  PMCE_TRY(pmce_opt_in_lds(reinterpret_cast<const void*>(&gemm_split_kernel<TM, TN, ACT, RES, APACK, OPACK, RS, KSUB, LNEP>), LDS, done,
                           "gemm_split_f16"));
  hipLaunchKernelGGL((gemm_split_kernel<TM, TN, ACT, RES, APACK, OPACK, RS, KSUB, LNEP>), dim3(grid), dim3(256), LDS, stream, p);
  return PMCE_OK;
}
// the 64x128 tile with two k-tiles per stage (small grids): the operand / epilogue combinations the model's products use
template <int KSUB>
static int launch_small_k32(SplitParams& p, int act, bool apack, bool opack, hipStream_t stream) {
  using Cfg = SplitCfg<1, 2>;
  p.ntm = (p.M + Cfg::BM - 1) / Cfg::BM;
  p.ntn = (p.N + Cfg::BN - 1) / Cfg::BN;
  int g = p.ntm * p.ntn;
  if (g > 512) g = 512;
  g = (g + 7) & ~7;
  const bool res = p.R != nullptr;
  if (p.rscale) return launch_one<1, 2, 0, false, true, false, true, KSUB>(p, g, stream);
  if (opack) return launch_one<1, 2, 1, false, true, true, false, KSUB>(p, g, stream);
  if (apack) return res ? launch_one<1, 2, 0, true, true, false, false, KSUB>(p, g, stream)
                        : launch_one<1, 2, 0, false, true, false, false, KSUB>(p, g, stream);
  return launch_one<1, 2, 0, false, false, false, false, KSUB>(p, g, stream);
}
static bool small_k32_form_exists(int act, bool apack, bool opack, bool res, bool rs) {
  if (rs || opack) return true;
  if (apack) return act == 0;
  return act == 0 && !res;
}
template <int TM, int TN>
static int launch_cfg(SplitParams& p, int act, bool apack, bool opack, hipStream_t stream) {
  using Cfg = SplitCfg<TM, TN>;
  p.ntm = (p.M + Cfg::BM - 1) / Cfg::BM;
  p.ntn = (p.N + Cfg::BN - 1) / Cfg::BN;
  const int per_cu = (160 * 1024) / Cfg::LDS_BYTES >= 3 ? 3 : 2;
  int g = p.ntm * p.ntn;
  if (g > 256 * per_cu) g = 256 * per_cu;
  g = (g + 7) & ~7;
  const bool res = p.R != nullptr;
  if (p.rscale) return launch_one<TM, TN, 0, false, true, false, true>(p, g, stream);  // raw-input products: packed, row-scaled A
  if (opack) return launch_one<TM, TN, 1, false, true, true>(p, g, stream);  // fc1 of the lifter: GELU, packed in, packed out
  if (apack) {
    if (act == 1) return res ? launch_one<TM, TN, 1, true, true>(p, g, stream) : launch_one<TM, TN, 1, false, true>(p, g, stream);
    return res ? launch_one<TM, TN, 0, true, true>(p, g, stream) : launch_one<TM, TN, 0, false, true>(p, g, stream);
  }
  if (act == 1) return res ? launch_one<TM, TN, 1, true, false>(p, g, stream) : launch_one<TM, TN, 1, false, false>(p, g, stream);
  return res ? launch_one<TM, TN, 0, true, false>(p, g, stream) : launch_one<TM, TN, 0, false, false>(p, g, stream);
}

// Tile choice: a launch takes as long as its busiest CU (tiles are spread evenly over an XCD's 32 CUs, two or three
// workgroups per CU sharing one matrix pipe); `ovh` is the per-area handicap of the smaller wave tiles (operand traffic and
// split work per matrix instruction).
struct SplitTile { int bm, bn; double ovh; };
static const SplitTile kSplitTiles[] = {{128, 256, 1.0}, {128, 128, 1.12}, {64, 128, 1.3}};
static int pick_split_tile(int M, int N) {
  const int forced = g_split_tile.load(std::memory_order_relaxed);
  if (forced >= 0 && forced < 3) return forced;
  // Small grids (fewer 128x256 tiles than CUs): the launch is ONE round of workgroups, each bound by its own serial chain per
  // k-tile (wait -> barrier -> DMA issue -> fragment reads -> 24 / 12 / 6 dependent matrix instructions per wave): 0.9 / 0.66 / 0.43 us
  // per iteration for the three shapes whatever the number of tiles (profiles/r04_c_*: the AdaLN product's 24 workgroups iterate as
  // fast as the GRU projection's 216).  Take the shape with the shortest chain whose tiles still run in one round (two workgroups per
  // CU interleave, three of the smallest shape fit).
  if ((long long)((M + 127) / 128) * ((N + 255) / 256) < 256) {
    static const double t_iter[3] = {0.9, 0.66, 0.43};
    static const int slots[3] = {512, 512, 768};
    int best = 0;
    double best_cost = 1e300;
    for (int i = 0; i < 3; ++i) {
      const SplitTile& t = kSplitTiles[i];
      const long long tiles = (long long)((M + t.bm - 1) / t.bm) * ((N + t.bn - 1) / t.bn);
      const double cost = (double)((tiles + slots[i] - 1) / slots[i]) * t_iter[i];
      if (cost < best_cost - 1e-12) { best_cost = cost; best = i; }
    }
    return best;
  }
  int best = 0;
  double best_cost = 1e300;
  for (int i = 0; i < 3; ++i) {
    const SplitTile& t = kSplitTiles[i];
    const long long tiles = (long long)((M + t.bm - 1) / t.bm) * ((N + t.bn - 1) / t.bn);
    const long long per_cu = ((tiles + 7) / 8 + 31) / 32;
    const double cost = (double)per_cu * t.bm * t.bn * t.ovh;
    if (cost < best_cost) { best_cost = cost; best = i; }
  }
  return best;
}

static int gemm_split_any(const float* A, const float* Wp, const float* wscale, const float* bias, const float* R, float* C, int M,
                          int N, int K, long long lda, long long ldc, int act, int a_packed, int c_packed, int c_div,
                          long long c_lo, long long c_hi, hipStream_t stream, const float* rscale = nullptr, int w_blocked = 0) {
  PMCE_REQUIRE(A && Wp && wscale && C, "gemm_split: null pointer");
  PMCE_REQUIRE(M > 0 && N > 0 && K >= 32 && K % 16 == 0, "gemm_split: need M,N>0, K>=32 and K%%16==0 (got M=%d N=%d K=%d)", M, N, K);
  PMCE_REQUIRE(act == 0 || act == 1, "gemm_split: act must be 0 or 1");
  PMCE_REQUIRE(lda >= K && lda % 4 == 0 && ldc >= N, "gemm_split: lda=%lld (>=K, multiple of 4) / ldc=%lld (>=N)", lda, ldc);
  PMCE_REQUIRE(!a_packed || lda == K, "gemm_split: a packed A has lda == K");
  PMCE_REQUIRE((long long)M * lda * 4 < (1ll << 32) && ((long long)N + 63) / 64 * 64 * K * 4 < (1ll << 32) && (long long)M * ldc < (1ll << 32),
               "gemm_split: an operand spans 4 GiB or more (split the batch)");
  PMCE_REQUIRE(ldc < (1ll << 21), "gemm_split: ldc=%lld (the tile epilogue addresses a 256-row window of C / R with 32-bit byte offsets)", ldc);
  PMCE_REQUIRE(c_div == 0 || R == nullptr, "gemm_split: a C row map cannot be combined with a residual");
  PMCE_REQUIRE(!c_packed || (a_packed && act == 1 && R == nullptr && c_div == 0 && N % 32 == 0 && ldc == N),
               "gemm_split: a packed result is supported for the packed-A + GELU form with N %% 32 == 0 and ldc == N");
  PMCE_REQUIRE(!rscale || (a_packed && !c_packed && act == 0 && R == nullptr && K >= 128),
               "gemm_split: a row-scaled A is a packed A with K >= 128; no activation, residual or packed result");
  // (K >= 128 is the RS epilogue's invariant: it reads the tile's row-scale slice from LDS at the END of the tile, while the DMA cursor runs
  // NS - 1 <= 3 stages ahead across tile boundaries - a tile must span at least NS - 1 stages (K / (16 KSUB) >= 3 with KSUB <= 2), or tile t + 2's
  // slice would land on tile t's scales before its epilogue.  ADVICE r04.)
  SplitParams p;
  p.A = A; p.W = Wp; p.wscale = wscale; p.bias = bias; p.R = R; p.C = C; p.rscale = rscale; p.wblk = w_blocked;
  p.M = M; p.N = N; p.K = K; p.lda = (unsigned)lda; p.ldc = (unsigned)ldc;
  p.c_div = c_div; p.c_lo = c_lo; p.c_hi = c_hi;
  p.ln1_w = p.ln1_b = p.ln2_w = p.ln2_b = nullptr; p.ln1_eps = p.ln2_eps = 0.f; p.out1 = p.out2 = nullptr;
  p.oflow = pmce_overflow_sink();
  p.clk = pmce_clock_sink();  // (thread-local: set while an entry point of a model with a clock probe runs)
  const int tile = pick_split_tile(M, N);
  // at most one round of 64 x 128 tiles: the small-grid kernel (gemm_split_small.hip: eight waves per tile, a deep ring, no tile stream)
  if (tile == 2 && g_split_tile.load(std::memory_order_relaxed) < 0 &&
      pmce_gemm_split_small_applies(M, N, K, act, a_packed != 0, c_packed != 0, R != nullptr, rscale != nullptr)) {
    PMCE_TRY(pmce_gemm_split_small_launch(p, act, a_packed != 0, c_packed != 0, stream));
    return pmce_check_launch("gemm_nt_split_f16 (small grid)");
  }
  if (tile == 2 && g_split_tile.load(std::memory_order_relaxed) < 0 && K % 32 == 0 && K >= 128 &&
      (long long)((M + 63) / 64) * ((N + 127) / 128) <= 512 &&
      small_k32_form_exists(act, a_packed != 0, c_packed != 0, R != nullptr, rscale != nullptr)) {
    PMCE_TRY(launch_small_k32<2>(p, act, a_packed != 0, c_packed != 0, stream));  // small grid: two k-tiles per trip
    return pmce_check_launch("gemm_nt_split_f16 (64x128, two k-tiles per stage)");
  }
  switch (tile) {
    case 0: PMCE_TRY((launch_cfg<2, 4>(p, act, a_packed != 0, c_packed != 0, stream))); break;
    case 1: PMCE_TRY((launch_cfg<2, 2>(p, act, a_packed != 0, c_packed != 0, stream))); break;
    default: PMCE_TRY((launch_cfg<1, 2>(p, act, a_packed != 0, c_packed != 0, stream))); break;
  }
  return pmce_check_launch("gemm_nt_split_f16");
}

extern "C" int pmce_gemm_nt_split_f16(const float* A, const float* Wp, const float* wscale, const float* bias, const float* R,
                                      float* C, int M, int N, int K, long long lda, long long ldc, int act, int a_packed,
                                      hipStream_t stream) {
  return gemm_split_any(A, Wp, wscale, bias, R, C, M, N, K, lda, ldc, act, a_packed, 0, 0, 0, 0, stream);
}
extern "C" int pmce_gemm_nt_split_f16_ex(const float* A, const float* Wp, const float* wscale, const float* bias, const float* R,
                                         float* C, int M, int N, int K, long long lda, long long ldc, int act, int a_packed,
                                         int c_packed, hipStream_t stream) {
  return gemm_split_any(A, Wp, wscale, bias, R, C, M, N, K, lda, ldc, act, a_packed, c_packed, 0, 0, 0, stream);
}
extern "C" int pmce_gemm_nt_split_f16_rowmap(const float* A, const float* Wp, const float* wscale, const float* bias, float* C,
                                             int M, int N, int K, long long lda, int c_div, long long c_lo, long long c_hi,
                                             hipStream_t stream) {
  PMCE_REQUIRE(c_div > 0, "gemm_split_rowmap: c_div must be positive");
  return gemm_split_any(A, Wp, wscale, bias, nullptr, C, M, N, K, lda, N, 0, 0, 0, c_div, c_lo, c_hi, stream);
}

// Every form above on a weight in the BLOCKED layout (pmce_gemm_pack_split_f16_blk): rscale != null -> A is row-scaled (as _rs),
// c_div > 0 -> mapped output rows (as _rowmap; ldc == N then), else as _ex.  What the model's launch sequences call.
extern "C" int pmce_gemm_nt_split_f16_blk(const float* A, const float* rscale, const float* Wblk, const float* wscale, const float* bias,
                                          const float* R, float* C, int M, int N, int K, long long lda, long long ldc, int act,
                                          int a_packed, int c_packed, int c_div, long long c_lo, long long c_hi, hipStream_t stream) {
  PMCE_REQUIRE(c_div >= 0 && (c_div == 0 || ldc == N), "gemm_split_blk: a row map needs ldc == N");
  return gemm_split_any(A, Wblk, wscale, bias, R, C, M, N, K, lda, ldc, act, rscale ? 1 : a_packed, c_packed, c_div, c_lo, c_hi, stream,
                        rscale, 1);
}

// Raw inputs (any finite fp32 magnitude): Ap / rscale from pmce_split_rows_scaled_f16.  C[m][n] = 2^e(m) 2^-s(n) (Ahi Whi + Ahi Wlo +
// Alo Whi) + bias[n]; c_div > 0 maps the output rows like pmce_gemm_nt_split_f16_rowmap (ldc = N then).
extern "C" int pmce_gemm_nt_split_f16_rs(const float* Ap, const float* rscale, const float* Wp, const float* wscale, const float* bias,
                                         float* C, int M, int N, int K, long long ldc, int c_div, long long c_lo, long long c_hi,
                                         hipStream_t stream) {
  PMCE_REQUIRE(rscale, "gemm_split_rs: null rscale");
  PMCE_REQUIRE(c_div >= 0 && (c_div == 0 || ldc == N), "gemm_split_rs: a row map needs ldc == N");
  return gemm_split_any(Ap, Wp, wscale, bias, nullptr, C, M, N, K, K, ldc, 0, 1, 0, c_div, c_lo, c_hi, stream, rscale);
}

// A product with N = 256 and a residual whose result feeds a LayerNorm chain (the proj and fc2 products of a C = 256 lifter block,
// reference PoseEstimation.py:26-28,84-85,91-92,101-106): with x = Ap W^T + bias + R,
//   y1 = ln1_w ? LN(x; ln1_w, ln1_b, ln1_eps) : x ;   out1 = y1 (fp32 [M,256], may alias R; may be null)
//   out2 = LN(y1; ln2_w, ln2_b, ln2_eps) written PRE-SPLIT (the next product's A; may be null)
// - pmce_gemm_nt_split_f16_blk followed by pmce_ln_chain_ex_f32(out2_split = 1) in one launch (64 x 256 tiles: a workgroup owns whole
// rows).  Ap pre-split [M,K]; W packed (w_blocked: the blocked layout); statistics two-pass in fp32 like pmce_ln_chain (a different
// summation order: the results agree to the last bits, not bit for bit).
extern "C" int pmce_gemm_nt_split_f16_ln(const float* Ap, const float* Wp, int w_blocked, const float* wscale, const float* bias,
                                         const float* R, int M, int K, const float* ln1_w, const float* ln1_b, float ln1_eps, float* out1,
                                         const float* ln2_w, const float* ln2_b, float ln2_eps, float* out2, hipStream_t stream) {
  PMCE_REQUIRE(Ap && Wp && wscale && R && (out1 || out2), "gemm_split_ln: null pointer");
  PMCE_REQUIRE(M > 0 && K >= 32 && K % 16 == 0, "gemm_split_ln: need M>0, K>=32 and K%%16==0 (got M=%d K=%d)", M, K);
  PMCE_REQUIRE((!ln1_w) == (!ln1_b) && (!out2 || (ln2_w && ln2_b)), "gemm_split_ln: a LayerNorm needs weight and bias");
  PMCE_REQUIRE((long long)M * K * 4 < (1ll << 32) && (long long)M * 256 * 4 < (1ll << 32), "gemm_split_ln: an operand spans 4 GiB or more (split the batch)");
  SplitParams p;
  p.A = Ap; p.W = Wp; p.wscale = wscale; p.bias = bias; p.R = R; p.C = nullptr; p.rscale = nullptr; p.wblk = w_blocked;
  p.M = M; p.N = 256; p.K = K; p.lda = (unsigned)K; p.ldc = 256u;
  p.c_div = 0; p.c_lo = 0; p.c_hi = 0;
  p.oflow = pmce_overflow_sink();
  p.clk = pmce_clock_sink();
  p.ln1_w = ln1_w; p.ln1_b = ln1_b; p.ln1_eps = ln1_eps; p.out1 = out1;
  p.ln2_w = ln2_w; p.ln2_b = ln2_b; p.ln2_eps = ln2_eps; p.out2 = out2;
  p.ntm = (M + 63) / 64;
  p.ntn = 1;
  int g = p.ntm > 512 ? 512 : p.ntm;
  g = (g + 7) & ~7;
  // (two k-tiles per stage on small grids, as the plain 64 x 128 tile has them, were measured and change nothing here: at M = 272 the
  // launch is its prologue and the epilogue's dependent chain, not the 16 trips of its k-loop)
  PMCE_TRY((launch_one<1, 4, 0, true, true, false, false, 1, true>(p, g, stream)));
  return pmce_check_launch("gemm_nt_split_f16_ln");
}

// ---- operand packing ------------------------------------------------------------------------------------------------------
// W[N][ldw] fp32 -> Wp[N][K/16][2][16] f16 (N*K floats of storage) and wscale[N] = 2^-s(n), s(n) chosen per row so that the
// row's max|W| * 2^s lies in [2^14, 2^15): hi = rne16(W 2^s), lo = rne16(W 2^s - hi) - both normal f16 for every weight within
// 2^-17 of the row's largest (smaller ones lose bits that are 2^-28 of the row's largest product).  One wave per row.
__global__ __launch_bounds__(256) void split_pack_kernel(const float* __restrict__ W, int N, int K, int ldw, _Float16* __restrict__ Wp,
                                                         float* __restrict__ wscale, int blocked) {
  const int lane = threadIdx.x & 63;
  for (long long n = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); n < N; n += (long long)gridDim.x * 4) {
    const float* __restrict__ row = W + n * ldw;
    float m = 0.f;
    for (int k = lane; k < K; k += 64) m = fmaxf(m, fabsf(row[k]));
    m = wave_max(m);
    int e = 0;
    if (m > 0.f && m < 3.0e38f) frexpf(m, &e);  // m = f * 2^e, f in [0.5, 1)
    e = max(-110, min(e, 125));                  // keeps 2^(15-e) and 2^(e-15) normal fp32 numbers
    const float up = m > 0.f ? ldexpf(1.f, 15 - e) : 1.f, down = m > 0.f ? ldexpf(1.f, e - 15) : 1.f;
    if (lane == 0) wscale[n] = down;
    // row-major: 32 f16 per (row, k-tile), a row's k-tiles adjacent; blocked: [n / 64][k-tile][n % 64][32 f16]
    _Float16* __restrict__ out = blocked ? Wp + ((n >> 6) * (long long)(K / 16) * 64 + (n & 63)) * 32 : Wp + n * K * 2;
    const long long kt_stride = blocked ? 64 * 32 : 32;
    for (int k = lane; k < K; k += 64) {
      const float ws = row[k] * up;
      const _Float16 hi = (_Float16)ws;
      const _Float16 lo = (_Float16)(ws - (float)hi);
      out[(k / 16) * kt_stride + (k % 16)] = hi;
      out[(k / 16) * kt_stride + 16 + (k % 16)] = lo;
    }
  }
}
extern "C" int pmce_gemm_pack_split_f16(const float* W, int N, int K, int ldw, float* Wp, float* wscale, hipStream_t stream) {
  PMCE_REQUIRE(W && Wp && wscale, "gemm_pack_split: null pointer");
  PMCE_REQUIRE(N > 0 && K > 0 && K % 16 == 0 && ldw >= K, "gemm_pack_split: need N>0, K%%16==0, ldw>=K (N=%d K=%d ldw=%d)", N, K, ldw);
  const int blocks = (N + 3) / 4 < 4096 ? (N + 3) / 4 : 4096;
  hipLaunchKernelGGL(split_pack_kernel, dim3(blocks), dim3(256), 0, stream, W, N, K, ldw, reinterpret_cast<_Float16*>(Wp), wscale, 0);
  return pmce_check_launch("gemm_pack_split_f16");
}
// The same planes in the BLOCKED layout [ceil(N/64)][K/16][64][16 hi | 16 lo] (Wp: ceil(N/64)*64*K floats of storage; rows past N are
// not written and never contribute): what a tile fetches per k-tile is then contiguous 4 KB pieces.  A streamed weight (one that is
// read once per launch from HBM: the final product's 278 MB, the GRU and AdaLN projections) otherwise arrives as 64-byte pieces one
// weight row apart - a different DRAM page per piece.
extern "C" int pmce_gemm_pack_split_f16_blk(const float* W, int N, int K, int ldw, float* Wp, float* wscale, hipStream_t stream) {
  PMCE_REQUIRE(W && Wp && wscale, "gemm_pack_split_blk: null pointer");
  PMCE_REQUIRE(N > 0 && K > 0 && K % 16 == 0 && ldw >= K, "gemm_pack_split_blk: need N>0, K%%16==0, ldw>=K (N=%d K=%d ldw=%d)", N, K, ldw);
  const int blocks = (N + 3) / 4 < 4096 ? (N + 3) / 4 : 4096;
  hipLaunchKernelGGL(split_pack_kernel, dim3(blocks), dim3(256), 0, stream, W, N, K, ldw, reinterpret_cast<_Float16*>(Wp), wscale, 1);
  return pmce_check_launch("gemm_pack_split_f16_blk");
}

// A[M][lda] fp32 -> Ap[M][K/16][2][16] f16: hi = rne16(a), lo = rne16((a - hi) * 2^11)
__global__ void split_rows_kernel(const float* __restrict__ A, long long M, int K, long long lda, _Float16* __restrict__ Ap) {
  const long long total = M * K;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long m = i / K;
    const int k = (int)(i % K);
    const float a = A[m * lda + k];
    const _Float16 hi = (_Float16)a;
    const _Float16 lo = (_Float16)((a - (float)hi) * 2048.0f);
    _Float16* row = Ap + (m * K + (k / 16) * 16) * 2;
    row[k % 16] = hi;
    row[16 + k % 16] = lo;
  }
}
extern "C" int pmce_split_rows_f16(const float* A, long long M, int K, long long lda, float* Ap, hipStream_t stream) {
  PMCE_REQUIRE(A && Ap && M > 0 && K > 0 && K % 16 == 0 && lda >= K, "split_rows: bad arguments");
  const long long total = M * K;
  const long long want = (total + 256 * 4 - 1) / (256 * 4);
  hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)(want < 8192 ? want : 8192)), dim3(256), 0, stream, A, M, K, lda,
                     reinterpret_cast<_Float16*>(Ap));
  return pmce_check_launch("split_rows_f16");
}

// Raw input rows of ANY finite fp32 magnitude -> packed planes of A[m] * 2^-e(m) plus rscale[m] = 2^e(m): e(m) puts the row's
// max |a| into [2^14, 2^15) (exactly what pmce_gemm_pack_split_f16 does for a weight row), so neither a feature of 1e5 (beyond f16's
// 65504) nor one of 1e-7 (whose planes would be f16 sub-normals) loses anything: both planes are normal f16 numbers for every
// element within 2^-17 of its row's largest.  The reference's Linear accepts any fp32 row (PoseEstimation.py:80,
// CoevoDecoder.py:228); so does this.  Rows holding inf / nan keep scale 1 and propagate as non-finite results of their own row
// only, like the reference's.  One wave per row; K % 16 == 0.
__global__ __launch_bounds__(256) void split_rows_scaled_kernel(const float* __restrict__ A, long long M, int K, long long lda,
                                                                _Float16* __restrict__ Ap, float* __restrict__ rscale) {
  const int lane = threadIdx.x & 63;
  for (long long m = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); m < M; m += (long long)gridDim.x * 4) {
    const float* __restrict__ row = A + m * lda;
    float mx = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(row + k);
      mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    mx = wave_max(mx);  // (fmaxf drops NaNs: a row with a NaN and finite values is scaled by its finite maximum, the NaN stays a NaN)
    int e = 0;
    if (mx > 0.f && mx < 3.0e38f) frexpf(mx, &e);  // mx = f * 2^e, f in [0.5, 1)
    e = max(-110, min(e, 125));
    const bool scaled = mx > 0.f && mx < 3.0e38f;
    const float down = scaled ? ldexpf(1.f, 15 - e) : 1.f, up = scaled ? ldexpf(1.f, e - 15) : 1.f;
    if (lane == 0) rscale[m] = up;
    _Float16* __restrict__ out = Ap + m * K * 2;
    for (int k = lane * 4; k < K; k += 256) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(row + k);
      const float a[4] = {pinned(v.x * down), pinned(v.y * down), pinned(v.z * down), pinned(v.w * down)};  // ONE fp32 value for both planes
      f16x4 hi, lo;
#pragma unroll
      for (int i = 0; i < 4; ++i) hi[i] = (_Float16)a[i];
#pragma unroll
      for (int i = 0; i < 4; ++i) lo[i] = (_Float16)((a[i] - (float)hi[i]) * 2048.0f);
      _Float16* q = out + (k >> 4) * 32 + (k & 15);
      *reinterpret_cast<f16x4*>(q) = hi;
      *reinterpret_cast<f16x4*>(q + 16) = lo;
    }
  }
}
extern "C" int pmce_split_rows_scaled_f16(const float* A, long long M, int K, long long lda, float* Ap, float* rscale, hipStream_t stream) {
  PMCE_REQUIRE(A && Ap && rscale && M > 0 && K > 0 && K % 16 == 0 && lda >= K && lda % 4 == 0, "split_rows_scaled: bad arguments");
  const long long want = (M + 3) / 4;
  hipLaunchKernelGGL(split_rows_scaled_kernel, dim3((unsigned)(want < 16384 ? want : 16384)), dim3(256), 0, stream, A, M, K, lda,
                     reinterpret_cast<_Float16*>(Ap), rscale);
  return pmce_check_launch("split_rows_scaled_f16");
}
