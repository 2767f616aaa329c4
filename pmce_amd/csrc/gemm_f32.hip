// fp32 MFMA "NT" GEMM for gfx950:  C[m,n] = epi( sum_k A[m,k] * W[n,k] + bias[n] )  (+ R[m,n])
//
// Every dense contraction of the path whose operands do not stay in registers goes through this kernel:
// the lifter's qkv / proj / fc1 / fc2 Linear layers (reference PoseEstimation.py:13-29 via timm
// Attention/Mlp), imgfeat_embed (PoseEstimation.py:80), the GRU input and recurrent projections
// (CoevoDecoder.py:216-221), the 24 live AdaLN gamma/beta Linear(2048->64) layers packed as one
// [B,2048]x[2048,3072] product (CoevoDecoder.py:19-20), and the final 431->6890 upsample conv + 3 residual
// Linear(2048->6890) packed as one [B,3360]x[3360,20670] product (CoevoDecoder.py:238-244).
//
// Design: v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD), 256 threads = 2x2 waves,
// block tile BMxBNx32, both operands K-contiguous so both tiles are staged with 128-byte-line global loads
// (8 lanes x 16 B per row), register-prefetched one tile ahead, double-buffered in LDS with a 36-float row
// stride (conflict-free ds_read_b128).  The k order inside each group of 8 is permuted (lanes 0-31 take
// k..k+3, lanes 32-63 take k+4..k+7) so that one ds_read_b128 per operand feeds four MFMAs.
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
};

template <int BM, int BN, int ACT, bool RES, bool CMAP>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmParams p) {
  constexpr int LD = 36;
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int LA = BM / 32, LB = BN / 32;  // float4 loads per thread per tile
  __shared__ __attribute__((aligned(16))) float As[2][BM * LD];
  __shared__ __attribute__((aligned(16))) float Bs[2][BN * LD];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int n0 = lane & 31, hb = lane >> 5;
  const int wm = wave & 1, wn = wave >> 1;

  // ---- XCD-aware, grouped tile order: consecutive tile ids of one XCD share A/W panels in its L2 ----
  const int nblk = p.ntm * p.ntn;
  int bid = blockIdx.x;
  {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;  // bijective remap
  }
  constexpr int GROUP_M = 8;
  const int per_group = GROUP_M * p.ntn;
  const int group = bid / per_group;
  const int first_m = group * GROUP_M;
  const int gsz = min(p.ntm - first_m, GROUP_M);
  const int tile_m = first_m + (bid % per_group) % gsz;
  const int tile_n = (bid % per_group) / gsz;
  const int m_base = tile_m * BM, n_base = tile_n * BN;

  const float* __restrict__ A = p.A + (long long)blockIdx.z * p.bsA;
  const float* __restrict__ W = p.W + (long long)blockIdx.z * p.bsW;

  // ---- per-thread global source pointers (fixed rows, advancing k).  Rows past the edge are CLAMPED to the
  // last valid row (their products are never stored), so the k-loop has no predicated loads or branches. ----
  const int kc = (tid & 7) * 4, r0 = tid >> 3;
  const float* aptr[LA];
  const float* bptr[LB];
#pragma unroll
  for (int i = 0; i < LA; ++i) {
    const int r = min(m_base + r0 + 32 * i, p.M - 1);
    aptr[i] = A + (long long)(r % p.a_div) * p.a_lo + (long long)(r / p.a_div) * p.a_hi + kc;
  }
#pragma unroll
  for (int i = 0; i < LB; ++i) {
    const int r = min(n_base + r0 + 32 * i, p.N - 1);
    bptr[i] = W + (long long)r * p.ldw + kc;
  }

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

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = p.K / 32;
  gload(0);
  lstore(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload(kt + 1);
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
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) lstore(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: bias, activation, residual; D layout: col = lane&31, row = (r&3)+8*(r>>2)+4*hb.
  // Interior tiles take a branch-free path: the 16 residual loads of a 32x32 sub-tile are issued back to back,
  // then the 16 stores (each a 128-byte row segment per half-wave). ----
  float* __restrict__ C = p.C + (long long)blockIdx.z * p.bsC;
  const float* __restrict__ R = RES ? p.R + (long long)blockIdx.z * p.bsC : nullptr;
  const float* __restrict__ bias = p.bias ? p.bias + (long long)blockIdx.z * p.bsBias : nullptr;
  const bool full = (m_base + BM <= p.M) && (n_base + BN <= p.N);
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n_base + wn * WN + j * 32 + n0;
    const bool nok = n < p.N;
    const float bv = (bias && nok) ? bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int mrow = m_base + wm * WM + i * 32 + 4 * hb;
      if (CMAP) {  // mapped rows (GRU layer-0 input projection: (b,t) rows -> time-major); never with act/residual
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mrow + (r & 3) + 8 * (r >> 2);
          if (nok && m < p.M)
            C[(long long)(m % p.c_div) * p.c_lo + (long long)(m / p.c_div) * p.c_hi + n] = acc[i][j][r] + bv;
        }
      } else {
        const int ldc = (int)p.c_lo;
        float* __restrict__ Cp = C + (long long)mrow * ldc + n;   // 32-bit offsets from here on
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
}

template <int BM, int BN>
static void launch_gemm(const GemmParams& p, int batch, bool cmap, hipStream_t stream) {
  const dim3 grid(p.ntm * p.ntn, 1, batch), block(256);
  const bool res = p.R != nullptr;
  if (cmap) {
    hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, 0, false, true>), grid, block, 0, stream, p);
  } else if (p.act == 1) {
    if (res) hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, 1, true, false>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, 1, false, false>), grid, block, 0, stream, p);
  } else {
    if (res) hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, 0, true, false>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, 0, false, false>), grid, block, 0, stream, p);
  }
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
  // tile choice: big tiles when they still fill the chip (>= 1 block per CU), else 64x64
  const long long big = (long long)((M + 127) / 128) * ((N + 127) / 128) * batch;
  if (big >= 256) {
    p.ntm = (M + 127) / 128; p.ntn = (N + 127) / 128;
    launch_gemm<128, 128>(p, batch, cmap, stream);
  } else {
    p.ntm = (M + 63) / 64; p.ntn = (N + 63) / 64;
    launch_gemm<64, 64>(p, batch, cmap, stream);
  }
  return pmce_check_launch("gemm_nt_f32");
}
