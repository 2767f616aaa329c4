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
// ------------------------------------------------------------------------------------------------------
struct GruStepArgs {
  const float* gi[2];     // + row*gi_rs + gate*H + unit   (b_ih already added by the input-projection GEMM)
  const float* whh[2];    // [3H][H]
  const float* bhh[2];    // [3H]
  const float* hprev[2];  // + row*h_rs + unit ; null -> h = 0 (first step)
  float* hout[2];         // + row*h_rs + unit
  long long gi_rs, h_rs;
  int B, H;
};

// NQ = number of K-slices (wave groups) per workgroup: 2 -> 4 waves, 4 -> 8 waves (two per SIMD, so one wave's LDS / barrier
// latency hides under the other's MFMAs).
template <int NQ>
__global__ __launch_bounds__(128 * NQ) void gru_step_kernel(GruStepArgs a) {
  constexpr int NT = 128 * NQ;        // threads
  constexpr int KH = 64 / NQ;         // k per K-slice per iteration (row of 64 floats + 4 pad in LDS)
  constexpr int LD = 68;              // conflict-free ds_read_b128 row stride
  constexpr int RW = 16 / NQ;         // accumulator rows each wave finishes
  constexpr int LA = 1024 / NT, LW = 1536 / NT;  // float4 loads per thread per iteration (A: 64x16, W: 96x16)
  __shared__ __attribute__((aligned(16))) float smem[2 * (64 + 96) * LD];
  const int d = blockIdx.z;
  const int u0 = blockIdx.x * 32, m0 = blockIdx.y * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
    // staging assignment per iteration: A 64 rows x 16 float4, W 96 rows x 16 float4; float4 c4 of a row belongs to
    // K-slice c4 / (16/NQ)
    constexpr int CPS = 16 / NQ;  // float4 per slice per row
    const float* ap[LA];
    const float* wp[LW];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const int idx = tid + NT * i, row = idx >> 4, c4 = idx & 15;
      const int m = min(m0 + row, a.B - 1);
      ap[i] = hp + (long long)m * a.h_rs + (c4 / CPS) * Kq + (c4 % CPS) * 4;
    }
#pragma unroll
    for (int i = 0; i < LW; ++i) {
      const int idx = tid + NT * i, row = idx >> 4, c4 = idx & 15;
      const int gate = row >> 5, uu = row & 31;
      wp[i] = W + (long long)(gate * H + u0 + uu) * H + (c4 / CPS) * Kq + (c4 % CPS) * 4;
    }
    f32x4 ra[LA], rw[LW];
    auto gload = [&](int kt) {
#pragma unroll
      for (int i = 0; i < LA; ++i) ra[i] = *reinterpret_cast<const f32x4*>(ap[i] + kt * KH);
#pragma unroll
      for (int i = 0; i < LW; ++i) rw[i] = *reinterpret_cast<const f32x4*>(wp[i] + kt * KH);
    };
    auto lstore = [&](int buf) {
      float* As = smem + buf * (64 + 96) * LD;
      float* Bs = As + 64 * LD;
#pragma unroll
      for (int i = 0; i < LA; ++i) {
        const int idx = tid + NT * i;
        *reinterpret_cast<f32x4*>(As + (idx >> 4) * LD + (idx & 15) * 4) = ra[i];
      }
#pragma unroll
      for (int i = 0; i < LW; ++i) {
        const int idx = tid + NT * i;
        *reinterpret_cast<f32x4*>(Bs + (idx >> 4) * LD + (idx & 15) * 4) = rw[i];
      }
    };
    const int nk = Kq / KH;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int buf = kt & 1;
      if (kt + 1 < nk) gload(kt + 1);
      const float* As = smem + buf * (64 + 96) * LD + (rb * 32 + n0) * LD + kq * KH + 4 * hb;
      const float* Bs = smem + buf * (64 + 96) * LD + 64 * LD + n0 * LD + kq * KH + 4 * hb;
#pragma unroll
      for (int g8 = 0; g8 < KH / 8; ++g8) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(As + 8 * g8);
        f32x4 bv[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) bv[g] = *reinterpret_cast<const f32x4*>(Bs + g * 32 * LD + 8 * g8);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int g = 0; g < 3; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[g][s], acc[g], 0, 0, 0);
      }
      if (kt + 1 < nk) lstore(buf ^ 1);
      __syncthreads();
    }
    // meet the K-slices: every wave publishes the rows its partners finish, then sums the partners' copies of its own
    float* red = smem;  // [dst kq][src slot][rb][gate][q][lane]  (the operand tiles are dead now)
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

extern "C" int pmce_gru_step_f32(const float* gi0, const float* gi1, const float* whh0, const float* whh1, const float* bhh0,
                                 const float* bhh1, const float* hp0, const float* hp1, float* ho0, float* ho1,
                                 long long gi_rs, long long h_rs, int B, int H, int ndir, hipStream_t stream) {
  PMCE_REQUIRE(ndir == 1 || ndir == 2, "gru_step: ndir must be 1 or 2");
  PMCE_REQUIRE(gi0 && whh0 && bhh0 && ho0 && B > 0, "gru_step: null pointer");
  PMCE_REQUIRE(ndir == 1 || (gi1 && whh1 && bhh1 && ho1), "gru_step: second direction pointers missing");
  PMCE_REQUIRE(H > 0 && H % 256 == 0 && h_rs % 4 == 0, "gru_step: H must be a multiple of 256, h_rs of 4");
  GruStepArgs a;
  a.gi[0] = gi0; a.gi[1] = gi1; a.whh[0] = whh0; a.whh[1] = whh1; a.bhh[0] = bhh0; a.bhh[1] = bhh1;
  a.hprev[0] = hp0; a.hprev[1] = hp1; a.hout[0] = ho0; a.hout[1] = ho1;
  a.gi_rs = gi_rs; a.h_rs = h_rs; a.B = B; a.H = H;
  static const int nq = getenv("PMCE_GRU_NQ") ? atoi(getenv("PMCE_GRU_NQ")) : 2;   // tuning knob: K-slices per workgroup (2 and 4 measure the same)
  if (nq == 2)
    hipLaunchKernelGGL((gru_step_kernel<2>), dim3(H / 32, (B + 63) / 64, ndir), dim3(256), 0, stream, a);
  else
    hipLaunchKernelGGL((gru_step_kernel<4>), dim3(H / 32, (B + 63) / 64, ndir), dim3(512), 0, stream, a);
  return pmce_check_launch("gru_step");
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
