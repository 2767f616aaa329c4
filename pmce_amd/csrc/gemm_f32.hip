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
//    operands K-contiguous so both tiles are staged with 128-byte-line global loads (8 lanes x 16 B per row),
//    register-prefetched one k-tile ahead, double-buffered in LDS with a 36-float row stride (conflict-free
//    ds_read_b128).  The k order inside each group of 8 is permuted (lanes 0-31 take k..k+3, lanes 32-63 take
//    k+4..k+7) so that one ds_read_b128 per operand feeds four MFMAs.
//  * PERSISTENT workgroups walk an XCD-local chunk of the tile order; the next tile's first k-tile is fetched
//    under the current tile's last MFMAs (no exposed prologue).
//  * TWO accumulator sets: the finished tile's epilogue (bias / GELU / residual / stores) is cut into 8 slices that
//    ride inside the first 8 k-iterations of the NEXT tile, so stores, erf and residual loads issue in the shadow
//    of the matrix pipe.  With K = 256 (8 k-iterations per tile) this is what lifts the lifter's Linear layers
//    out of the regime where co-resident workgroups run their prologues and epilogues in lockstep.
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"

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

template <int TN>
struct PendingEpi {  // the finished-but-not-yet-stored tile
  float* c;          // &C[row0][col0] of this lane's first element
  const float* r;    // same for the residual
  float bv[TN];      // bias of this lane's column in each column tile
  bool valid;        // wave-uniform
};

template <int BM, int BN, int WGM, int ACT, bool RES, bool CMAP>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmParams p) {
  constexpr int LD = 36;
  constexpr int WGN = 4 / WGM;  // 4 waves arranged WGM x WGN
  constexpr int WM = BM / WGM, WN = BN / WGN;
  static_assert(WM % 32 == 0 && WN % 32 == 0, "wave tile must be a multiple of the 32x32 MFMA tile");
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int LA = BM / 32, LB = BN / 32;  // float4 loads per thread per tile
  constexpr int EPS = TM * TN * 2;           // accumulator elements per epilogue slice (8 slices per tile)
  __shared__ __attribute__((aligned(16))) float As[2][BM * LD];
  __shared__ __attribute__((aligned(16))) float Bs[2][BN * LD];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
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
  const int kc = (tid & 7) * 4, r0 = tid >> 3;
  const float* aptr[LA];
  const float* bptr[LB];
  auto set_ptrs = [&](int mb, int nb) {
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const int r = min(mb + r0 + 32 * i, p.M - 1);
      aptr[i] = A + (long long)(r % p.a_div) * p.a_lo + (long long)(r / p.a_div) * p.a_hi + kc;
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
      const int r = min(nb + r0 + 32 * i, p.N - 1);
      bptr[i] = W + (long long)r * p.ldw + kc;
    }
  };

  f32x4 ra[LA], rb[LB];
  auto gload = [&](int kt) {
    const int ko = kt * 32;
#pragma unroll
    for (int i = 0; i < LA; ++i) ra[i] = *reinterpret_cast<const f32x4*>(aptr[i] + ko);
#pragma unroll
    for (int i = 0; i < LB; ++i) rb[i] = *reinterpret_cast<const f32x4*>(bptr[i] + ko);
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < LA; ++i) *reinterpret_cast<f32x4*>(&As[buf][(r0 + 32 * i) * LD + kc]) = ra[i];
#pragma unroll
    for (int i = 0; i < LB; ++i) *reinterpret_cast<f32x4*>(&Bs[buf][(r0 + 32 * i) * LD + kc]) = rb[i];
  };

  const int nk = p.K / 32;
  const bool pipelined = !CMAP && nk >= 8;  // the sliced epilogue needs 8 k-iterations to ride in
  int li = bx;
  if (li >= chunk_len) return;
  int m_base, n_base, m_next = 0, n_next = 0;
  tile_coords(chunk_start + li, m_base, n_base);
  set_ptrs(m_base, n_base);
  gload(0);
  lstore(0);
  __syncthreads();
  int buf = 0;
  bool has_next = false;

  // Element e of a wave tile (e = slice*EPS + u): accumulator register r = e % 16 of sub-tile ij = e / 16
  // (i = ij % TM, j = ij / TM); its offset from the lane's first element is (i*32 + (r&3) + 8*(r>>2))*ldc + j*32.
#define EPI_OFF(e) ((((((e) / 16) % TM) * 32 + (((e) % 16) & 3) + 8 * (((e) % 16) >> 2)) * ldc) + (((e) / 16) / TM) * 32)

  // one k-iteration into `cur`; if S >= 0 it also carries slice S of the pending tile's (`prv`) epilogue
  auto iteration = [&](f32x16(&cur)[TM][TN], const f32x16(&prv)[TM][TN], PendingEpi<TN>& pend, int kt, auto slice_tag) {
    constexpr int S = decltype(slice_tag)::value;
    constexpr int SS = S < 0 ? 0 : S;
    bool loaded = true;
    if (kt + 1 < nk) {
      gload(kt + 1);
    } else if (has_next) {  // cross-tile prefetch: the next tile's first k-tile flies under this tile's last MFMAs
      set_ptrs(m_next, n_next);
      gload(0);
    } else {
      loaded = false;
    }
    float rv[EPS];
    if (S >= 0 && RES && pend.valid) {
#pragma unroll
      for (int u = 0; u < EPS; ++u) rv[u] = pend.r[EPI_OFF(SS * EPS + u)];
    }
    const float* as = &As[buf][(wm * WM + n0) * LD + 4 * hb];
    const float* bs = &Bs[buf][(wn * WN + n0) * LD + 4 * hb];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(as + i * 32 * LD + 8 * g);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4*>(bs + j * 32 * LD + 8 * g);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            cur[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], cur[i][j], 0, 0, 0);
    }
    if (S >= 0 && pend.valid) {
#pragma unroll
      for (int u = 0; u < EPS; ++u) {
        const int e = SS * EPS + u;
        float v = prv[(e / 16) % TM][(e / 16) / TM][e % 16] + pend.bv[(e / 16) / TM];
        if (ACT == 1) v = gelu_erf(v);
        if (RES) v += rv[u];
        pend.c[EPI_OFF(e)] = v;
      }
    }
    if (loaded) lstore(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  };

  // k-loop of one tile into `cur`, with the pending tile's epilogue slices riding in iterations 0..7
  auto run_tile = [&](f32x16(&cur)[TM][TN], const f32x16(&prv)[TM][TN], PendingEpi<TN>& pend) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) cur[i][j][r] = 0.f;
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
      const float bv = (bias && nok) ? bias[n] : 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int mrow = mb + wm * WM + i * 32 + 4 * hb;
        if (CMAP) {  // mapped rows (GRU layer-0 input projection: (b,t) rows -> time-major); never with act/residual
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = mrow + (r & 3) + 8 * (r >> 2);
            if (nok && m < p.M)
              C[(long long)(m % p.c_div) * p.c_lo + (long long)(m / p.c_div) * p.c_hi + n] = acc[i][j][r] + bv;
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
              float v = acc[i][j][r] + bv;
              if (ACT == 1) v = gelu_erf(v);
              if (RES) v += rv[r];
              Cp[((r & 3) + 8 * (r >> 2)) * ldc] = v;
            }
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int rr = (r & 3) + 8 * (r >> 2);
              if (nok && mrow + rr < p.M) {
                float v = acc[i][j][r] + bv;
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
  auto finish_tile = [&](const f32x16(&acc)[TM][TN], PendingEpi<TN>& pend) {
    const bool full = (m_base + BM <= p.M) && (n_base + BN <= p.N);
    if (pipelined && full && has_next) {
      const long long o = (long long)(m_base + wm * WM + 4 * hb) * ldc + n_base + wn * WN + n0;
      pend.c = C + o;
      pend.r = RES ? R + o : nullptr;
#pragma unroll
      for (int j = 0; j < TN; ++j) pend.bv[j] = bias ? bias[n_base + wn * WN + j * 32 + n0] : 0.f;
      pend.valid = true;
    } else {
      epilogue_now(acc, m_base, n_base);
      pend.valid = false;
    }
  };

  f32x16 acc0[TM][TN], acc1[TM][TN];
  PendingEpi<TN> pend;
  pend.valid = false;
  pend.c = nullptr;
  pend.r = nullptr;
#pragma unroll
  for (int j = 0; j < TN; ++j) pend.bv[j] = 0.f;

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

// Tile choice.  All tiles of a launch take the same time, so a launch runs in ceil(tiles / resident slots) rounds;
// M = B*16*J is rarely a multiple of 128*256 (B=256, J=17: 544 row tiles = 2.125 x 256 CUs), so the tile shape is
// picked to minimise rounds * (blocks per CU) * tile area.  Blocks per CU are bounded by LDS (2 x (BM+BN) x 144 B).
struct TileCfg { int bm, bn, bpc; };
static const TileCfg kTiles[] = {{128, 128, 2}, {96, 128, 2}, {64, 128, 2}, {64, 64, 4}};
static int pick_tile(int M, int N, int batch) {
  if (const char* e = getenv("PMCE_GEMM_TILE")) {  // tuning/debug knob: force a tile config (0..3)
    const int v = atoi(e);
    if (v >= 0 && v < 4) return v;
  }
  int best = 0;
  double best_cost = 1e300;
  for (int i = 0; i < 4; ++i) {
    const TileCfg& t = kTiles[i];
    const long long tiles = (long long)((M + t.bm - 1) / t.bm) * ((N + t.bn - 1) / t.bn) * batch;
    const long long slots = 256ll * t.bpc;
    const long long rounds = (tiles + slots - 1) / slots;
    // a partially filled single round only costs what its busiest CU runs
    const long long per_cu = rounds > 1 ? rounds * t.bpc : (tiles + 255) / 256;
    const double cost = (double)per_cu * t.bm * t.bn * (1.0 + 2048.0 / (t.bm * t.bn));  // mild bias to big tiles
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
  p.act = act;
  p.bsA = bsA; p.bsW = bsW; p.bsBias = bsBias; p.bsC = bsC;
  const int ti = pick_tile(M, N, batch);
  p.ntm = (M + kTiles[ti].bm - 1) / kTiles[ti].bm;
  p.ntn = (N + kTiles[ti].bn - 1) / kTiles[ti].bn;
  p.grid_cap = 256 * kTiles[ti].bpc;
  if (const char* e = getenv("PMCE_GEMM_GRID")) {  // tuning knob: persistent workgroups per CU
    const int v = atoi(e);
    if (v >= 1 && v <= 8) p.grid_cap = 256 * v;
  }
  switch (ti) {
    case 0: launch_gemm<128, 128, 2>(p, batch, cmap, stream); break;
    case 1: launch_gemm<96, 128, 1>(p, batch, cmap, stream); break;
    case 2: launch_gemm<64, 128, 2>(p, batch, cmap, stream); break;
    default: launch_gemm<64, 64, 2>(p, batch, cmap, stream); break;
  }
  return pmce_check_launch("gemm_nt_f32");
}
