// Entry points of scripts/microbench/libpmce_diag.so - the diagnostics library.  NOT part of the product: libpmce_hip.so ships one
// split-f16 GEMM (pmce_amd/csrc/gemm_split_f16.hip) and the fp32 one; the two experimental variants kept here for the record
// (DESIGN.md, appendix on the split GEMM's ceiling) are
//   kind 0: the wave-specialised 192x256 kernel (gemm_split_ws.hip: correct, not faster),
//   kind 1: the same products on v_mfma_f32_16x16x32_f16 (gemm_split_m16.hip: +6-14 % isolated on the wide product, nothing in the model),
// and dbg_victims.hip holds the self-checking bystander kernels of the matrix-pipe interference report
// (tests/test_gpu_bystander.py, scripts/microbench/victims.py).  Same operand layout and arithmetic as pmce_gemm_nt_split_f16_ex.
#include "gemm_split_common.hpp"

bool pmce_gemm_split_ws_wants(int M, int N, int K, int a_packed, int c_div);
int pmce_gemm_split_ws_launch(SplitParams& p, int act, int c_packed, hipStream_t stream);
bool pmce_gemm_split_m16_wants(int K, int a_packed, int c_div);
int pmce_gemm_split_m16_launch(SplitParams& p, int act, int c_packed, int tile, hipStream_t stream);

// tile: 0 = 128x256, 1 = 128x128, 2 = 64x128 (kind 1 only).  Returns PMCE_ERR_ARG when the variant does not apply to the product.
extern "C" int pmce_diag_gemm_nt_split_f16(int kind, const float* A, const float* Wp, const float* wscale, const float* bias,
                                           const float* R, float* C, int M, int N, int K, long long lda, long long ldc, int act,
                                           int a_packed, int c_packed, int tile, hipStream_t stream) {
  PMCE_REQUIRE(A && Wp && wscale && C, "diag gemm_split: null pointer");
  PMCE_REQUIRE(M > 0 && N > 0 && K >= 32 && K % 16 == 0 && lda >= K && ldc >= N, "diag gemm_split: bad shape");
  PMCE_REQUIRE(!c_packed || (a_packed && act == 1 && R == nullptr && N % 32 == 0 && ldc == N), "diag gemm_split: bad packed-result request");
  SplitParams p;
  p.A = A; p.W = Wp; p.wscale = wscale; p.bias = bias; p.R = R; p.C = C; p.rscale = nullptr; p.wblk = 0;
  p.M = M; p.N = N; p.K = K; p.lda = (unsigned)lda; p.ldc = (unsigned)ldc;
  p.c_div = 0; p.c_lo = 0; p.c_hi = 0;
  p.ln1_w = p.ln1_b = p.ln2_w = p.ln2_b = nullptr; p.ln1_eps = p.ln2_eps = 0.f; p.out1 = p.out2 = nullptr;
  p.oflow = nullptr;
  p.clk = nullptr;
  p.skew = (K / 16) * 12 * 32 / 4096 + 1;
  if (kind == 0) {
    PMCE_REQUIRE(pmce_gemm_split_ws_wants(M, N, K, a_packed, 0), "diag gemm_split: the wave-specialised kernel does not apply to this product");
    PMCE_TRY(pmce_gemm_split_ws_launch(p, act, c_packed, stream));
    return pmce_check_launch("diag gemm_split (ws)");
  }
  PMCE_REQUIRE(kind == 1, "diag gemm_split: kind must be 0 (wave-specialised) or 1 (16x16x32)");
  PMCE_REQUIRE(pmce_gemm_split_m16_wants(K, a_packed, 0), "diag gemm_split: the 16x16x32 kernel needs a pre-split A and K %% 32 == 0");
  PMCE_TRY(pmce_gemm_split_m16_launch(p, act, c_packed, tile, stream));
  return pmce_check_launch("diag gemm_split (16x16x32)");
}
