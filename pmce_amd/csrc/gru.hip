// Bidirectional 2-layer GRU gate update (reference CoevoDecoder.py:216-221,228: nn.GRU(2048,1024,bidirectional,
// num_layers=2), seq-first, h0 = 0).  The input and recurrent projections are gemm_f32.hip launches; this kernel
// is the point-wise part of one time step for up to two directions:
//   r = sigmoid(gi_r + gh_r) ; z = sigmoid(gi_z + gh_z) ; n = tanh(gi_n + r * gh_n) ; h' = (1 - z) * n + z * h
// with gi = W_ih x + b_ih and gh = W_hh h + b_hh (PyTorch gate order r, z, n).
#include "common.hpp"

struct GruGateArgs {
  const float* gi[2];
  const float* gh[2];
  const float* hprev[2];  // null -> h = 0
  float* hout[2];
  long long gi_rs, gh_rs[2], hp_rs, ho_rs;  // row strides in floats (gh_rs = 0 broadcasts b_hh on the first step)
  int B, H;
};

__global__ __launch_bounds__(256) void gru_gates_kernel(GruGateArgs a) {
  const int d = blockIdx.y;
  const int h4 = a.H >> 2;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= a.B * h4) return;
  const int b = idx / h4, u = (idx % h4) * 4;
  const float* gi = a.gi[d] + (long long)b * a.gi_rs + u;
  const float* gh = a.gh[d] + (long long)b * a.gh_rs[d] + u;
  const f32x4 gir = *reinterpret_cast<const f32x4*>(gi), giz = *reinterpret_cast<const f32x4*>(gi + a.H),
              gin = *reinterpret_cast<const f32x4*>(gi + 2 * a.H);
  const f32x4 ghr = *reinterpret_cast<const f32x4*>(gh), ghz = *reinterpret_cast<const f32x4*>(gh + a.H),
              ghn = *reinterpret_cast<const f32x4*>(gh + 2 * a.H);
  f32x4 hp = {0.f, 0.f, 0.f, 0.f};
  if (a.hprev[d]) hp = *reinterpret_cast<const f32x4*>(a.hprev[d] + (long long)b * a.hp_rs + u);
  f32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float r = sigmoidf_acc(gir[i] + ghr[i]);
    const float z = sigmoidf_acc(giz[i] + ghz[i]);
    const float n = tanhf(gin[i] + r * ghn[i]);
    o[i] = (1.0f - z) * n + z * hp[i];
  }
  *reinterpret_cast<f32x4*>(a.hout[d] + (long long)b * a.ho_rs + u) = o;
}

extern "C" int pmce_gru_gates_f32(const float* gi0, const float* gi1, const float* gh0, const float* gh1, const float* hp0,
                                  const float* hp1, float* ho0, float* ho1, long long gi_rs, long long gh_rs0,
                                  long long gh_rs1, long long hp_rs, long long ho_rs, int B, int H, int ndir,
                                  hipStream_t stream) {
  PMCE_REQUIRE(ndir == 1 || ndir == 2, "gru_gates: ndir must be 1 or 2");
  PMCE_REQUIRE(gi0 && gh0 && ho0 && B > 0 && H > 0 && H % 4 == 0, "gru_gates: bad args");
  PMCE_REQUIRE(ndir == 1 || (gi1 && gh1 && ho1), "gru_gates: second direction pointers missing");
  GruGateArgs a;
  a.gi[0] = gi0; a.gi[1] = gi1; a.gh[0] = gh0; a.gh[1] = gh1;
  a.hprev[0] = hp0; a.hprev[1] = hp1; a.hout[0] = ho0; a.hout[1] = ho1;
  a.gi_rs = gi_rs; a.gh_rs[0] = gh_rs0; a.gh_rs[1] = gh_rs1; a.hp_rs = hp_rs; a.ho_rs = ho_rs;
  a.B = B; a.H = H;
  const int n = B * (H / 4);
  hipLaunchKernelGGL(gru_gates_kernel, dim3((n + 255) / 256, ndir), dim3(256), 0, stream, a);
  return pmce_check_launch("gru_gates");
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
