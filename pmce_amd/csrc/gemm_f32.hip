// fp32 MFMA "NT" GEMM for gfx950:  C[m,n] = epi( sum_k A[m,k] * W[n,k] + bias[n] )  (+ R[m,n])
//
// Every dense contraction of the path whose operands do not stay in registers goes through this kernel:
// the lifter's qkv / proj / fc1 / fc2 Linear layers (reference PoseEstimation.py:13-29 via timm
// Attention/Mlp), imgfeat_embed (PoseEstimation.py:80), the GRU input projections (CoevoDecoder.py:216-221),
// the 24 live AdaLN gamma/beta Linear(2048->64) layers packed as one [B,2048]x[2048,3072] product
// (CoevoDecoder.py:19-20), and the final 431->6890 upsample conv + 3 residual Linear(2048->6890) packed as one
// [B,3360]x[3360,20670] product (CoevoDecoder.py:238-244).
//
// Design:
//  * v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD), 256 threads = 4 waves, block tile BMxBNx32, both
//    operands K-contiguous.  Tiles go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds: 8 lanes x 16 B per 128-byte
//    row, no VGPR round trip, no ds_write), one k-tile ahead into a double-buffered LDS whose rows are unpadded with
//    their 16-byte chunks XOR-swizzled (swizzle on the DMA's source address and on the ds_read address: conflict-free
//    ds_read_b128).  The k order inside each group of 8 is permuted (lanes 0-31 take k..k+3, lanes 32-63 take
//    k+4..k+7) so that one ds_read_b128 per operand feeds four MFMAs.
//  * PERSISTENT workgroups walk an XCD-local chunk of the tile order; the next tile's first k-tile is fetched
//    under the current tile's last MFMAs (no exposed prologue).
//  * TWO accumulator sets: the finished tile's epilogue (GELU / residual / stores) is cut into 8 slices that ride
//    inside the first 8 k-iterations of the NEXT tile.  The bias is the accumulators' initial value; stores and
//    residual loads are buffer instructions (descriptor on the wave tile, scalar element offset, one per-lane offset
//    VGPR for the whole kernel), so a slice is one VMEM instruction (+ packed GELU) per element and no address arithmetic
//    on the vector unit, which shares its FMA lanes with the fp32 matrix pipe.  Stores are streaming (nt); residuals are
//    fetched two slices ahead (the residual stream is cold when a launch starts).
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"

#ifndef PMCE_ST_AUX
#define PMCE_ST_AUX 2  // cache policy of the epilogue stores: 2 = nt (streaming), 0 = default
#endif

struct GemmParams {
  const float* A;
  const float* W;
  const float* bias;  // [N] or null
  const float* R;     // residual, same row map as C, or null
  float* C;
  int M, N, K;
  int ldw;
  int a_div;  // A row r -> A + (r % a_div) * a_lo + (r / a_div) * a_hi   (elements)
  long long a_lo, a_hi;
  int c_div;  // C/R row r -> (r % c_div) * c_lo + (r / c_div) * c_hi
  long long c_lo, c_hi;
  int act;  // 0 none, 1 exact-erf GELU
  long long bsA, bsW, bsBias, bsC;  // per-blockIdx.z offsets (elements)
  int ntm, ntn;                     // tile counts
  int grid_cap;                     // persistent workgroups per batch entry
};

struct PendingEpi {  // the finished-but-not-yet-stored tile (all fields wave-uniform: they live in SGPRs)
  float* c;          // &C[first row of the wave tile][first column of the wave tile]
  const float* r;    // same for the residual
  bool valid;
};

// LDS-DMA: one wave instruction moves 64 x 16 B from per-lane buffer offsets straight into LDS at M0 + lane*16 (no
// VGPR round trip, no ds_write).  Buffer form: descriptor on the operand, a 32-bit per-lane byte offset that is fixed
// for a whole tile, the k-tile's byte offset in soffset - the k-loop spends no vector instruction on addresses.
// hipcc neither counts nor waits for it: the k-loop drains it with an explicit s_waitcnt vmcnt(0) ahead of its
// barrier.  M0 is compiler-reserved, hence saved/restored inside the statement.
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, int soff, unsigned lds_wave_base) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(rsrc), "s"(lds_wave_base), "s"(soff)
      : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const float* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)p;
}

template <int BM, int BN, int WGM, int ACT, bool RES, bool CMAP>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmParams p) {
  constexpr int LD = 32;  // unpadded rows (the DMA writes lane-linearly); 16-byte chunks XOR-swizzled instead
  constexpr int WGN = 4 / WGM;  // 4 waves arranged WGM x WGN
  constexpr int WM = BM / WGM, WN = BN / WGN;
  static_assert(WM % 32 == 0 && WN % 32 == 0, "wave tile must be a multiple of the 32x32 MFMA tile");
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int LA = BM / 32, LB = BN / 32;  // float4 loads per thread per tile
  constexpr int EPS = TM * TN * 2;           // accumulator elements per epilogue slice (8 slices per tile)
  __shared__ __attribute__((aligned(16))) float As[2][BM * LD];
  __shared__ __attribute__((aligned(16))) float Bs[2][BN * LD];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // SGPR: wave-tile origins stay scalar
  const int n0 = lane & 31, hb = lane >> 5;
  const int wm = wave % WGM, wn = wave / WGM;

  // ---- persistent workgroups.  The hardware places workgroup b on XCD b % 8; XCD x owns a contiguous chunk of
  // the (grouped) tile order and its resident workgroups walk that chunk round-robin, so the tiles in flight on
  // one XCD share A / W panels in its L2.  Correctness never depends on the placement. ----
  const int nblk = p.ntm * p.ntn;
  const int xcd = blockIdx.x & 7, bx = blockIdx.x >> 3, gx = gridDim.x >> 3;
  const int cq = nblk >> 3, cr = nblk & 7;
  const int chunk_start = xcd < cr ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq;
  const int chunk_len = cq + (xcd < cr ? 1 : 0);
  constexpr int GROUP_M = 8;
  const int per_group = GROUP_M * p.ntn;
  auto tile_coords = [&](int bid, int& mb, int& nb) {
    const int group = bid / per_group;
    const int first_m = group * GROUP_M;
    const int gsz = min(p.ntm - first_m, GROUP_M);
    mb = (first_m + (bid % per_group) % gsz) * BM;
    nb = ((bid % per_group) / gsz) * BN;
  };

  const float* __restrict__ A = p.A + (long long)blockIdx.z * p.bsA;
  const float* __restrict__ W = p.W + (long long)blockIdx.z * p.bsW;
  float* __restrict__ C = p.C + (long long)blockIdx.z * p.bsC;
  const float* __restrict__ R = RES ? p.R + (long long)blockIdx.z * p.bsC : nullptr;
  const float* __restrict__ bias = p.bias ? p.bias + (long long)blockIdx.z * p.bsBias : nullptr;
  const int ldc = (int)p.c_lo;

  // ---- per-thread global source pointers (fixed rows, advancing k).  Rows past the edge are CLAMPED to the
  // last valid row (their products are never stored), so the k-loop has no predicated loads or branches. ----
  // DMA instruction i of a wave fills tile rows 32*i + 8*wave .. +7: lane L lands at row (L>>3), PHYSICAL chunk (L&7).
  // Physical chunk p of tile row r holds logical chunk p ^ swz(r), swz(r) = (r>>1)&7 (the same involution on the
  // ds_read side): 16 consecutive lanes of a ds_read_b128 (rows n..n+15, one logical chunk) hit 16 distinct
  // 4-bank groups of the 64 banks.
  const int r0 = tid >> 3;
  const int kc = ((tid & 7) ^ ((r0 >> 1) & 7)) * 4;
  unsigned aoff[LA], boff[LB];  // byte offsets from A / W (the launcher checks that both operands span < 4 GiB)
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, 0xffffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, 0xffffffff, 0x00020000);
  // Both operands span < 4 GiB (checked by the launcher), so the byte offsets fit 32 bits: one 32-bit multiply-add per
  // row (the tile switch sits on the matrix pipe's critical path: every vector instruction there is matrix time lost).
  // The row-mapped form of A (a_div) keeps the general arithmetic; no product of the path uses it.
  const bool a_plain = p.a_div == 0x7fffffff;  // wave-uniform
  const unsigned lda_u = (unsigned)p.a_lo, ldw_u = (unsigned)p.ldw;
  auto set_ptrs = [&](int mb, int nb) {
    if (a_plain) {
#pragma unroll
      for (int i = 0; i < LA; ++i) aoff[i] = ((unsigned)min(mb + r0 + 32 * i, p.M - 1) * lda_u + (unsigned)kc) * 4u;
    } else {
#pragma unroll
      for (int i = 0; i < LA; ++i) {
        const int r = min(mb + r0 + 32 * i, p.M - 1);
        aoff[i] = (unsigned)(((long long)(r % p.a_div) * p.a_lo + (long long)(r / p.a_div) * p.a_hi + kc) * 4);
      }
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) boff[i] = ((unsigned)min(nb + r0 + 32 * i, p.N - 1) * ldw_u + (unsigned)kc) * 4u;
  };

  const unsigned lds_a = __builtin_amdgcn_readfirstlane(lds_addr(&As[0][0]) + wave * (8 * LD * 4));
  const unsigned lds_b = __builtin_amdgcn_readfirstlane(lds_addr(&Bs[0][0]) + wave * (8 * LD * 4));
  auto gdma = [&](int kt, int buf) {  // k-tile kt of the pointed-at tile -> LDS buffer `buf`
    const int ko = kt * 128;  // bytes
#pragma unroll
    for (int i = 0; i < LA; ++i) dma16(rsrc_a, aoff[i], ko, lds_a + (buf * BM + 32 * i) * (LD * 4));
#pragma unroll
    for (int i = 0; i < LB; ++i) dma16(rsrc_w, boff[i], ko, lds_b + (buf * BN + 32 * i) * (LD * 4));
  };
  auto dma_wait_and_sync = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };

  const int nk = p.K / 32;
  // the sliced epilogue needs 8 k-iterations to ride in (and a wave tile inside its descriptor's 2 GiB window)
  const bool pipelined = !CMAP && nk >= 8 && p.c_lo < (1ll << 21);
  int li = bx;
  if (li >= chunk_len) return;
  int m_base, n_base, m_next = 0, n_next = 0;
  tile_coords(chunk_start + li, m_base, n_base);
  set_ptrs(m_base, n_base);
  gdma(0, 0);
  dma_wait_and_sync();
  int buf = 0;
  bool has_next = false;

  // Element e of a wave tile (e = slice*EPS + u): accumulator register r = e % 16 of sub-tile ij = e / 16
  // (i = ij % TM, j = ij / TM).  Its address is  (wave-tile origin + ROW(e)*ldc + COL(e))  +  lane_off  with
  // lane_off = 4*hb*ldc + n0: a buffer descriptor on the wave-tile origin, the element's scalar offset in soffset and ONE
  // per-lane VGPR offset for the whole kernel (buffer_store_dword v, voff, s[rsrc], soff offen) - no address VALU at
  // all.  VALU cycles are matrix cycles here (§3.1), and 64-bit pointer arithmetic per store was most of the
  // epilogue's cost.
#define EPI_ROW(e) ((((e) / 16) % TM) * 32 + (((e) % 16) & 3) + 8 * (((e) % 16) >> 2))
#define EPI_COL(e) ((((e) / 16) / TM) * 32)
#define EPI_OFF(e) (EPI_ROW(e) * ldc + EPI_COL(e))
  // byte offset for the buffer forms.  `ldcb` is re-materialised (opaque to the optimiser) in every slice: otherwise all
  // 16*TM*TN loop-invariant soffsets are hoisted out of the k-loop, overflow the SGPR file and come back through
  // v_readlane + 5 wait states per store.  One s_mul per element on the otherwise idle scalar unit is free.
#define EPI_BYTES(e, ldcb) (EPI_ROW(e) * (ldcb) + EPI_COL(e) * 4)
  auto opaque_ldcb = [&]() {
    int v = ldc * 4;
    asm volatile("" : "+s"(v));
    return v;
  };
  const unsigned lane_off = (unsigned)(4 * hb * ldc + n0) * 4u;  // bytes
  constexpr int RSRC_FLAGS = 0x00020000;                          // raw dword buffer (gfx9 DATA_FORMAT = 32)

  // The bias is the accumulators' initial value (no add in the epilogue); the next tile's is fetched a tile ahead.
  float bv_next[TN];
  auto load_bias = [&](int nb) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = nb + wn * WN + j * 32 + n0;
      bv_next[j] = (bias && n < p.N) ? bias[n] : 0.f;
    }
  };

  // one k-iteration into `cur`; if S >= 0 it also carries slice S of the pending tile's (`prv`) epilogue.  Order:
  // slice S (its residuals were loaded an iteration ago), DMA of the next k-tile, residual loads of slice S+1, MFMAs.
  // Every vector-memory operation is thus issued ahead of the MFMAs and the vmcnt(0) in front of the barrier finds
  // them long done; and no compiler-counted load is consumed while an (uncounted) DMA is younger than it.
  // residuals, fetched TWO slices ahead (slice S uses rv[S & 1]; one k-iteration of a 64x64 tile is shorter than an HBM
  // round trip, and the residual stream is cold when the launch starts)
  float rv[2][EPS];
  const int swz = (n0 >> 1) & 7;
  auto iteration = [&](f32x16(&cur)[TM][TN], const f32x16(&prv)[TM][TN], PendingEpi& pend, int kt, auto slice_tag) {
    constexpr int S = decltype(slice_tag)::value;
    constexpr int SS = S < 0 ? 0 : S;
    if (S >= 0 && pend.valid) {
      const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(pend.c, 0, 0x7fffffff, RSRC_FLAGS);
      const int ldcb = opaque_ldcb();
#pragma unroll
      for (int u = 0; u < EPS; u += 2) {  // EPS is even; pairs share the packed GELU
        const int e = SS * EPS + u;
        f32x2 v = {prv[(e / 16) % TM][(e / 16) / TM][e % 16], prv[((e + 1) / 16) % TM][((e + 1) / 16) / TM][(e + 1) % 16]};
        if (ACT == 1) v = gelu_erf2(v);
        if (RES) v += f32x2{rv[SS & 1][u], rv[SS & 1][u + 1]};
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v.x), rc, lane_off, EPI_BYTES(e, ldcb), PMCE_ST_AUX);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v.y), rc, lane_off, EPI_BYTES(e + 1, ldcb), PMCE_ST_AUX);
      }
    }
    if (kt + 1 < nk) {
      gdma(kt + 1, buf ^ 1);
    } else if (has_next) {  // cross-tile prefetch: the next tile's first k-tile flies under this tile's last MFMAs
      set_ptrs(m_next, n_next);
      gdma(0, buf ^ 1);
    }
    if (S >= 0 && S < 6 && RES && pend.valid) {
      const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pend.r), 0, 0x7fffffff, RSRC_FLAGS);
      const int ldcb = opaque_ldcb();
#pragma unroll
      for (int u = 0; u < EPS; ++u)
        rv[SS & 1][u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rr, lane_off, EPI_BYTES((SS + 2) * EPS + u, ldcb), 0));
    }
    const float* as = &As[buf][(wm * WM + n0) * LD];
    const float* bs = &Bs[buf][(wn * WN + n0) * LD];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int co = 4 * ((2 * g + hb) ^ swz);  // physical position of logical chunk 2g+hb in this lane's rows
      f32x4 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(as + i * 32 * LD + co);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4*>(bs + j * 32 * LD + co);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            cur[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], cur[i][j], 0, 0, 0);
    }
    dma_wait_and_sync();
    buf ^= 1;
  };

  // k-loop of one tile into `cur`, with the pending tile's epilogue slices riding in iterations 0..7
  auto run_tile = [&](f32x16(&cur)[TM][TN], const f32x16(&prv)[TM][TN], PendingEpi& pend) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) cur[i][j][r] = bv_next[j];
    if (has_next) load_bias(n_next);
    if (pipelined) {  // nk >= 8: the first eight iterations are straight-line code, each with its own slice
      iteration(cur, prv, pend, 0, std::integral_constant<int, 0>{});
      iteration(cur, prv, pend, 1, std::integral_constant<int, 1>{});
      iteration(cur, prv, pend, 2, std::integral_constant<int, 2>{});
      iteration(cur, prv, pend, 3, std::integral_constant<int, 3>{});
      iteration(cur, prv, pend, 4, std::integral_constant<int, 4>{});
      iteration(cur, prv, pend, 5, std::integral_constant<int, 5>{});
      iteration(cur, prv, pend, 6, std::integral_constant<int, 6>{});
      iteration(cur, prv, pend, 7, std::integral_constant<int, 7>{});
      for (int kt = 8; kt < nk; ++kt) iteration(cur, prv, pend, kt, std::integral_constant<int, -1>{});
    } else {
      for (int kt = 0; kt < nk; ++kt) iteration(cur, prv, pend, kt, std::integral_constant<int, -1>{});
    }
    pend.valid = false;  // all 8 slices of the previous tile are out
  };

  // whole-tile epilogue (edge tiles, mapped rows, K < 256, and the last tile of a workgroup)
  auto epilogue_now = [&](const f32x16(&acc)[TM][TN], int mb, int nb) {
    const bool full = (mb + BM <= p.M) && (nb + BN <= p.N);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = nb + wn * WN + j * 32 + n0;
      const bool nok = n < p.N;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int mrow = mb + wm * WM + i * 32 + 4 * hb;
        if (CMAP) {  // mapped rows (GRU layer-0 input projection: (b,t) rows -> time-major); never with act/residual
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = mrow + (r & 3) + 8 * (r >> 2);
            if (nok && m < p.M)
              C[(long long)(m % p.c_div) * p.c_lo + (long long)(m / p.c_div) * p.c_hi + n] = acc[i][j][r];
          }
        } else {
          float* __restrict__ Cp = C + (long long)mrow * ldc + n;  // 32-bit offsets from here on
          const float* __restrict__ Rp = RES ? R + (long long)mrow * ldc + n : nullptr;
          if (full) {
            float rv[16];
            if (RES) {
#pragma unroll
              for (int r = 0; r < 16; ++r) rv[r] = Rp[((r & 3) + 8 * (r >> 2)) * ldc];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              float v = acc[i][j][r];
              if (ACT == 1) v = gelu_erf(v);
              if (RES) v += rv[r];
              Cp[((r & 3) + 8 * (r >> 2)) * ldc] = v;
            }
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int rr = (r & 3) + 8 * (r >> 2);
              if (nok && mrow + rr < p.M) {
                float v = acc[i][j][r];
                if (ACT == 1) v = gelu_erf(v);
                if (RES) v += Rp[rr * ldc];
                Cp[rr * ldc] = v;
              }
            }
          }
        }
      }
    }
  };

  // after a tile's k-loop: either park it as pending (interior tile with a successor) or store it now
  auto finish_tile = [&](const f32x16(&acc)[TM][TN], PendingEpi& pend) {
    const bool full = (m_base + BM <= p.M) && (n_base + BN <= p.N);
    if (pipelined && full && has_next) {
      const long long o = (long long)(m_base + wm * WM) * ldc + n_base + wn * WN;
      pend.c = C + o;
      pend.r = RES ? R + o : nullptr;
      if (RES) {
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pend.r), 0, 0x7fffffff, RSRC_FLAGS);
        const int ldcb = opaque_ldcb();
#pragma unroll
        for (int u = 0; u < 2 * EPS; ++u)   // slices 0 and 1
          rv[u / EPS][u % EPS] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rr, lane_off, EPI_BYTES(u, ldcb), 0));
      }
      pend.valid = true;
    } else {
      epilogue_now(acc, m_base, n_base);
      pend.valid = false;
    }
  };

  f32x16 acc0[TM][TN], acc1[TM][TN];
  PendingEpi pend;
  pend.valid = false;
  pend.c = nullptr;
  pend.r = nullptr;
  load_bias(n_base);

  while (true) {
    has_next = li + gx < chunk_len;
    if (has_next) tile_coords(chunk_start + li + gx, m_next, n_next);
    run_tile(acc0, acc1, pend);
    finish_tile(acc0, pend);
    if (!has_next) break;
    li += gx;
    m_base = m_next;
    n_base = n_next;

    has_next = li + gx < chunk_len;
    if (has_next) tile_coords(chunk_start + li + gx, m_next, n_next);
    run_tile(acc1, acc0, pend);
    finish_tile(acc1, pend);
    if (!has_next) break;
    li += gx;
    m_base = m_next;
    n_base = n_next;
  }
#undef EPI_OFF
#undef EPI_ROW
#undef EPI_COL
#undef EPI_BYTES
}

template <int BM, int BN, int WGM>
static void launch_gemm(const GemmParams& p, int batch, bool cmap, hipStream_t stream) {
  int g = p.ntm * p.ntn;
  if (g > p.grid_cap) g = p.grid_cap;
  g = (g + 7) & ~7;  // the XCD chunking wants a multiple of 8 workgroups
  const dim3 grid(g, 1, batch), block(256);
  const bool res = p.R != nullptr;
  if (cmap) {
    hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, WGM, 0, false, true>), grid, block, 0, stream, p);
  } else if (p.act == 1) {
    if (res) hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, WGM, 1, true, false>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, WGM, 1, false, false>), grid, block, 0, stream, p);
  } else {
    if (res) hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, WGM, 0, true, false>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, WGM, 0, false, false>), grid, block, 0, stream, p);
  }
}

// Tile choice.  Persistent workgroups spread an XCD's chunk of tiles evenly over its 32 CUs whatever the number of
// resident workgroups, and co-resident workgroups share one matrix pipe, so a launch takes as long as its busiest CU:
// ceil(ceil(tiles / 8) / 32) tiles of BM x BN.  M = B*16*J is rarely a multiple of 128*256 (B=256, J=17: 544 row tiles
// of 128), so the shape that quantises best wins; `ovh` is the measured per-area handicap of the smaller tiles at large
// K (twice the operand traffic per FLOP for 64x64).  Blocks per CU (LDS: 2 x (BM+BN) x 128 B; VGPRs) only size the grid.
// Tuning overrides (never set in production): changed only through pmce_gemm_set_tuning (scripts/gemm_sweep.py).
static std::atomic<int> g_force_tile{-1};
static std::atomic<int> g_force_grid{0};
extern "C" int pmce_gemm_set_tuning(int tile, int grid_per_cu) {
  g_force_tile.store(tile, std::memory_order_relaxed);
  g_force_grid.store(grid_per_cu, std::memory_order_relaxed);
  return PMCE_OK;
}

struct TileCfg { int bm, bn, bpc; double ovh; };
static const TileCfg kTiles[] = {{128, 128, 2, 1.0}, {96, 128, 2, 1.015}, {64, 128, 3, 1.03}, {64, 64, 4, 1.03}};
static int pick_tile(int M, int N, int batch) {
  const int forced = g_force_tile.load(std::memory_order_relaxed);  // tuning/debug knob: force a tile config (0..3)
  if (forced >= 0 && forced < 4) return forced;
  int best = 0;
  double best_cost = 1e300;
  for (int i = 0; i < 4; ++i) {
    const TileCfg& t = kTiles[i];
    const long long tiles = (long long)((M + t.bm - 1) / t.bm) * ((N + t.bn - 1) / t.bn) * batch;
    const long long per_cu = ((tiles + 7) / 8 + 31) / 32;
    const double cost = (double)per_cu * t.bm * t.bn * t.ovh;
    if (cost < best_cost) { best_cost = cost; best = i; }
  }
  return best;
}

extern "C" int pmce_gemm_nt_f32(const float* A, const float* W, const float* bias, const float* R, float* C, int M, int N,
                                int K, long long lda, int ldw, long long ldc, int act, int a_div, long long a_lo,
                                long long a_hi, int c_div, long long c_lo, long long c_hi, int batch, long long bsA,
                                long long bsW, long long bsBias, long long bsC, hipStream_t stream) {
  PMCE_REQUIRE(A && W && C, "gemm: null pointer");
  PMCE_REQUIRE(M > 0 && N > 0 && K > 0 && K % 32 == 0, "gemm: need M,N>0 and K%%32==0 (got M=%d N=%d K=%d)", M, N, K);
  PMCE_REQUIRE(ldw >= K && ldw % 4 == 0, "gemm: ldw=%d must be >=K and a multiple of 4", ldw);
  PMCE_REQUIRE(act == 0 || act == 1, "gemm: act must be 0 or 1");
  PMCE_REQUIRE(batch >= 1, "gemm: batch must be >= 1");
  GemmParams p;
  p.A = A; p.W = W; p.bias = bias; p.R = R; p.C = C;
  p.M = M; p.N = N; p.K = K; p.ldw = ldw;
  if (a_div <= 0) { p.a_div = 0x7fffffff; p.a_lo = lda; p.a_hi = 0; } else { p.a_div = a_div; p.a_lo = a_lo; p.a_hi = a_hi; }
  if (c_div <= 0) { p.c_div = 0x7fffffff; p.c_lo = ldc; p.c_hi = 0; } else { p.c_div = c_div; p.c_lo = c_lo; p.c_hi = c_hi; }
  PMCE_REQUIRE(p.a_lo % 4 == 0 && p.a_hi % 4 == 0, "gemm: A row strides must be multiples of 4 floats (16-byte loads)");
  const bool cmap = c_div > 0;
  PMCE_REQUIRE(!cmap || (act == 0 && R == nullptr), "gemm: a C row map cannot be combined with act/residual");
  PMCE_REQUIRE(cmap || p.c_lo < (1ll << 26), "gemm: ldc too large");
  {  // the LDS-DMA addresses operands with 32-bit byte offsets
    const long long am = M - 1;
    const long long a_span = (am < p.a_div ? am : (long long)p.a_div - 1) * p.a_lo + (am / p.a_div) * p.a_hi + K;  // upper bound
    PMCE_REQUIRE(a_span * 4 < (1ll << 32) && (long long)N * ldw * 4 < (1ll << 32),
                 "gemm: an operand spans 4 GiB or more per batch entry (split the batch)");
  }
  p.act = act;
  p.bsA = bsA; p.bsW = bsW; p.bsBias = bsBias; p.bsC = bsC;
  const int ti = pick_tile(M, N, batch);
  p.ntm = (M + kTiles[ti].bm - 1) / kTiles[ti].bm;
  p.ntn = (N + kTiles[ti].bn - 1) / kTiles[ti].bn;
  p.grid_cap = 256 * kTiles[ti].bpc;
  const int grid_knob = g_force_grid.load(std::memory_order_relaxed);  // tuning knob: persistent workgroups per CU
  if (grid_knob >= 1 && grid_knob <= 8) p.grid_cap = 256 * grid_knob;
  switch (ti) {
    case 0: launch_gemm<128, 128, 2>(p, batch, cmap, stream); break;
    case 1: launch_gemm<96, 128, 1>(p, batch, cmap, stream); break;
    case 2: launch_gemm<64, 128, 2>(p, batch, cmap, stream); break;
    default: launch_gemm<64, 64, 2>(p, batch, cmap, stream); break;
  }
  return pmce_check_launch("gemm_nt_f32");
}
