// On-device evaluation metrics directly behind the hot path: per-sample MPVPE / MPJPE / PA-MPJPE
// and the acceleration error, replacing the reference's per-batch D2H copy of [B,6890,3] meshes + numpy
// (data/PW3D/dataset.py:269-282 compute_both_err, :351-462 evaluate; lib/coord_utils.py:151-173 rigid_align,
// :218-245 compute_error_accel).  The mesh part is a 165 KB/sample streaming reduction (HBM-bound); the 14-joint
// Procrustes alignment (3x3 SVD) runs in fp64 on one lane per sample.
#include "common.hpp"

#define MAXJ 32

// cyclic Jacobi eigen-decomposition of a symmetric 3x3 matrix (fp64): S = V diag(w) V^T, columns of V = eigenvectors
__device__ void jacobi3(double S[3][3], double V[3][3], double w[3]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 24; ++sweep) {
    const double off = S[0][1] * S[0][1] + S[0][2] * S[0][2] + S[1][2] * S[1][2];
    if (off < 1e-300) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (fabs(S[p][q]) < 1e-300) continue;
        const double theta = (S[q][q] - S[p][p]) / (2.0 * S[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) {  // S <- S J
          const double skp = S[k][p], skq = S[k][q];
          S[k][p] = c * skp - s * skq;
          S[k][q] = s * skp + c * skq;
        }
        for (int k = 0; k < 3; ++k) {  // S <- J^T S
          const double spk = S[p][k], sqk = S[q][k];
          S[p][k] = c * spk - s * sqk;
          S[q][k] = s * spk + c * sqk;
        }
        for (int k = 0; k < 3; ++k) {  // V <- V J
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq;
          V[k][q] = s * vkp + c * vkq;
        }
      }
  }
  for (int i = 0; i < 3; ++i) w[i] = S[i][i];
}

// One workgroup per sample.
//   mesh error:  mean_v || (pm*scale - rp) - (gm*scale - rg) ||                         (dataset.py:271,279 / :389)
//   joints:      P = pj - rowsum*rp (if rowsum) ; P -= P[root_j] ; P = P[eval_idx]      (dataset.py:272,277 / :392-397)
//   MPJPE = mean_j ||P - G|| ; PA-MPJPE = mean_j ||rigid_align(P, G) - G||               (:431-433, coord_utils.py:151-173)
// rp / rg == nullptr  ->  roots are the samples' own joint root_j (compute_both_err semantics).
// V == 0 (pm / gm unused, may be null)  ->  joints only: the pose-only flavours (compute_joint_err / evaluate_joint, Human36M/dataset.py:600-713,
// PW3D/dataset.py:260-349; MPII3D.evaluate, MPII3D/dataset.py:539-624, whose compute_both_err reports a mesh error of 0); out_mpvpe = 0.
__global__ __launch_bounds__(256) void sample_errors_kernel(const float* __restrict__ pm, const float* __restrict__ gm,
                                                            float scale, int V, const float* __restrict__ rp,
                                                            const float* __restrict__ rg, const float* __restrict__ pj,
                                                            const float* __restrict__ gj, int NJ,
                                                            const float* __restrict__ rowsum, const int* __restrict__ eval_idx,
                                                            int n_eval, int root_j, float* __restrict__ out_mpvpe,
                                                            float* __restrict__ out_mpjpe, float* __restrict__ out_pampjpe,
                                                            float* __restrict__ out_pe, float* __restrict__ out_ge) {
  __shared__ double red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* pjb = pj + (long long)b * NJ * 3;
  const float* gjb = gj + (long long)b * NJ * 3;
  float rpx, rpy, rpz, rgx, rgy, rgz;
  if (rp) {
    rpx = rp[b * 3]; rpy = rp[b * 3 + 1]; rpz = rp[b * 3 + 2];
    rgx = rg[b * 3]; rgy = rg[b * 3 + 1]; rgz = rg[b * 3 + 2];
  } else {
    rpx = pjb[root_j * 3]; rpy = pjb[root_j * 3 + 1]; rpz = pjb[root_j * 3 + 2];
    rgx = gjb[root_j * 3]; rgy = gjb[root_j * 3 + 1]; rgz = gjb[root_j * 3 + 2];
  }
  // ---- mesh part: streaming reduction ----
  const float* pmb = pm + (long long)b * V * 3;
  const float* gmb = gm + (long long)b * V * 3;
  float acc = 0.f;
  for (int v = tid; v < V; v += 256) {
    const float dx = (pmb[v * 3] * scale - rpx) - (gmb[v * 3] * scale - rgx);
    const float dy = (pmb[v * 3 + 1] * scale - rpy) - (gmb[v * 3 + 1] * scale - rgy);
    const float dz = (pmb[v * 3 + 2] * scale - rpz) - (gmb[v * 3 + 2] * scale - rgz);
    acc += sqrtf(dx * dx + dy * dy + dz * dz);
  }
  double s = (double)wave_sum(acc);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  if (tid != 0) return;
  out_mpvpe[b] = V > 0 ? (float)((red[0] + red[1] + red[2] + red[3]) / V) : 0.f;

  // ---- joints: one lane, fp64 ----
  double P[MAXJ][3], G[MAXJ][3];
  double prt[3], grt[3];
  for (int k = 0; k < 3; ++k) {
    const double rs = rowsum ? (double)rowsum[root_j] : 0.0;
    prt[k] = (double)pjb[root_j * 3 + k] - rs * (k == 0 ? rpx : k == 1 ? rpy : rpz);
    grt[k] = (double)gjb[root_j * 3 + k] - rs * (k == 0 ? rgx : k == 1 ? rgy : rgz);
  }
  for (int e = 0; e < n_eval; ++e) {
    const int j = eval_idx[e];
    const double rs = rowsum ? (double)rowsum[j] : 0.0;
    for (int k = 0; k < 3; ++k) {
      P[e][k] = ((double)pjb[j * 3 + k] - rs * (k == 0 ? rpx : k == 1 ? rpy : rpz)) - prt[k];
      G[e][k] = ((double)gjb[j * 3 + k] - rs * (k == 0 ? rgx : k == 1 ? rgy : rgz)) - grt[k];
    }
  }
  double mp = 0.0;
  for (int e = 0; e < n_eval; ++e) {
    const double dx = P[e][0] - G[e][0], dy = P[e][1] - G[e][1], dz = P[e][2] - G[e][2];
    mp += sqrt(dx * dx + dy * dy + dz * dz);
    if (out_pe) {
      for (int k = 0; k < 3; ++k) {
        out_pe[((long long)b * n_eval + e) * 3 + k] = (float)P[e][k];
        out_ge[((long long)b * n_eval + e) * 3 + k] = (float)G[e][k];
      }
    }
  }
  out_mpjpe[b] = (float)(mp / n_eval);
  // ---- Procrustes (coord_utils.py:151-167) ----
  double cA[3] = {0, 0, 0}, cB[3] = {0, 0, 0};
  for (int e = 0; e < n_eval; ++e)
    for (int k = 0; k < 3; ++k) {
      cA[k] += P[e][k];
      cB[k] += G[e][k];
    }
  for (int k = 0; k < 3; ++k) {
    cA[k] /= n_eval;
    cB[k] /= n_eval;
  }
  double H[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  double varP = 0.0;
  for (int e = 0; e < n_eval; ++e)
    for (int i = 0; i < 3; ++i) {
      const double a = P[e][i] - cA[i];
      varP += a * a;
      for (int j = 0; j < 3; ++j) H[i][j] += a * (G[e][j] - cB[j]);
    }
  varP /= n_eval;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) H[i][j] /= n_eval;
  // SVD of H through the eigen-decomposition of H^T H:  H = U diag(sv) Vm^T
  double S[3][3], Vm[3][3], w[3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) S[i][j] = H[0][i] * H[0][j] + H[1][i] * H[1][j] + H[2][i] * H[2][j];
  jacobi3(S, Vm, w);
  int ord[3] = {0, 1, 2};  // descending eigenvalues
  for (int i = 0; i < 2; ++i)
    for (int j = i + 1; j < 3; ++j)
      if (w[ord[j]] > w[ord[i]]) {
        const int t = ord[i];
        ord[i] = ord[j];
        ord[j] = t;
      }
  double sv[3], U[3][3], Vs[3][3];
  for (int c = 0; c < 3; ++c) {
    sv[c] = sqrt(fmax(w[ord[c]], 0.0));
    for (int i = 0; i < 3; ++i) Vs[i][c] = Vm[i][ord[c]];
  }
  for (int c = 0; c < 2; ++c) {
    const double inv = sv[c] > 1e-300 ? 1.0 / sv[c] : 0.0;
    for (int i = 0; i < 3; ++i) U[i][c] = (H[i][0] * Vs[0][c] + H[i][1] * Vs[1][c] + H[i][2] * Vs[2][c]) * inv;
  }
  if (sv[2] > 1e-12 * sv[0]) {
    for (int i = 0; i < 3; ++i) U[i][2] = (H[i][0] * Vs[0][2] + H[i][1] * Vs[1][2] + H[i][2] * Vs[2][2]) / sv[2];
  } else {  // rank-deficient: complete the basis
    U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
    U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
    U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
  }
  double Rm[3][3];
  auto build_R = [&]() {  // R = V U^T
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) Rm[i][j] = Vs[i][0] * U[j][0] + Vs[i][1] * U[j][1] + Vs[i][2] * U[j][2];
  };
  build_R();
  const double det = Rm[0][0] * (Rm[1][1] * Rm[2][2] - Rm[1][2] * Rm[2][1]) - Rm[0][1] * (Rm[1][0] * Rm[2][2] - Rm[1][2] * Rm[2][0]) +
                     Rm[0][2] * (Rm[1][0] * Rm[2][1] - Rm[1][1] * Rm[2][0]);
  if (det < 0) {  // reflection: flip the last singular pair (coord_utils.py:158-161)
    sv[2] = -sv[2];
    for (int i = 0; i < 3; ++i) Vs[i][2] = -Vs[i][2];
    build_R();
  }
  const double cs = (sv[0] + sv[1] + sv[2]) / varP;
  double tt[3];
  for (int i = 0; i < 3; ++i) tt[i] = -cs * (Rm[i][0] * cA[0] + Rm[i][1] * cA[1] + Rm[i][2] * cA[2]) + cB[i];
  double pa = 0.0;
  for (int e = 0; e < n_eval; ++e) {
    double d2 = 0.0;
    for (int i = 0; i < 3; ++i) {
      const double a = cs * (Rm[i][0] * P[e][0] + Rm[i][1] * P[e][1] + Rm[i][2] * P[e][2]) + tt[i] - G[e][i];
      d2 += a * a;
    }
    pa += sqrt(d2);
  }
  out_pampjpe[b] = (float)(pa / n_eval);
}

// acceleration error per sample (coord_utils.py:218-245 as used by dataset.py:415-429): the first and last sample of
// every sequence contribute 0; acc[n] = mean_j || (P[n-1]-2P[n]+P[n+1]) - (G[n-1]-2G[n]+G[n+1]) ||.
__global__ __launch_bounds__(256) void accel_error_kernel(const float* __restrict__ pe, const float* __restrict__ ge,
                                                          const int* __restrict__ seq, float* __restrict__ out, int N,
                                                          int n_eval) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float r = 0.f;
  if (n > 0 && n + 1 < N && seq[n - 1] == seq[n] && seq[n + 1] == seq[n]) {
    double acc = 0.0;
    for (int e = 0; e < n_eval; ++e) {
      double d2 = 0.0;
      for (int k = 0; k < 3; ++k) {
        const long long o = ((long long)n * n_eval + e) * 3 + k, st = (long long)n_eval * 3;
        const double ap = (double)pe[o - st] - 2.0 * (double)pe[o] + (double)pe[o + st];
        const double ag = (double)ge[o - st] - 2.0 * (double)ge[o] + (double)ge[o + st];
        d2 += (ap - ag) * (ap - ag);
      }
      acc += sqrt(d2);
    }
    r = (float)(acc / n_eval);
  }
  out[n] = r;
}

extern "C" int pmce_sample_errors_f32(const float* pm, const float* gm, float scale, int V, const float* rp, const float* rg,
                                      const float* pj, const float* gj, int NJ, const float* rowsum, const int* eval_idx,
                                      int n_eval, int root_j, float* out_mpvpe, float* out_mpjpe, float* out_pampjpe,
                                      float* out_pe, float* out_ge, int B, hipStream_t stream) {
  PMCE_REQUIRE(((pm && gm) || V == 0) && pj && gj && eval_idx && out_mpvpe && out_mpjpe && out_pampjpe, "sample_errors: null pointer");
  PMCE_REQUIRE((rp == nullptr) == (rg == nullptr), "sample_errors: give both mesh roots or neither");
  PMCE_REQUIRE((out_pe == nullptr) == (out_ge == nullptr), "sample_errors: give both joint outputs or neither");
  PMCE_REQUIRE(B > 0 && V >= 0 && NJ > 0 && NJ <= MAXJ && n_eval >= 3 && n_eval <= MAXJ && root_j >= 0 && root_j < NJ,
               "sample_errors: bad sizes (NJ, n_eval <= 32; n_eval >= 3; V = 0 for joints only)");
  PMCE_REQUIRE(V > 0 || (rp == nullptr && rowsum == nullptr), "sample_errors: joints only (V = 0) takes no mesh roots and no rowsum");
  hipLaunchKernelGGL(sample_errors_kernel, dim3(B), dim3(256), 0, stream, pm, gm, scale, V, rp, rg, pj, gj, NJ, rowsum,
                     eval_idx, n_eval, root_j, out_mpvpe, out_mpjpe, out_pampjpe, out_pe, out_ge);
  return pmce_check_launch("sample_errors");
}

extern "C" int pmce_accel_error_f32(const float* pe, const float* ge, const int* seq, float* out, int N, int n_eval,
                                    hipStream_t stream) {
  PMCE_REQUIRE(pe && ge && seq && out && N > 0 && n_eval > 0, "accel_error: bad args");
  hipLaunchKernelGGL(accel_error_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, pe, ge, seq, out, N, n_eval);
  return pmce_check_launch("accel_error");
}

// ------------------------------------------------------------------------------------------------------
// Window assembly for sliding-window evaluation (lib/_img_utils.py:42-55; demo lib/utils/_dataset_demo.py:98-102):
// out_feat[w][t][:] = feat[frame(w,t)][:], out_pose likewise, frame(w,t) = start + t, or start when start == end.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void assemble_windows_kernel(const float* __restrict__ pose, const float* __restrict__ feat,
                                                               const int* __restrict__ win, float* __restrict__ out_pose,
                                                               float* __restrict__ out_feat, int W, int L, int J) {
  const int wt = blockIdx.x;  // (window, t)
  const int w = wt >> 4, t = wt & 15;
  const int s = win[2 * w], e = win[2 * w + 1];
  int fr = (s == e) ? s : s + t;
  fr = min(max(fr, 0), L - 1);
  const f32x4* src = reinterpret_cast<const f32x4*>(feat + (long long)fr * 2048);
  f32x4* dst = reinterpret_cast<f32x4*>(out_feat + (long long)wt * 2048);
  for (int i = threadIdx.x; i < 512; i += 256) dst[i] = src[i];
  const float* ps = pose + (long long)fr * J * 2;
  float* pd = out_pose + (long long)wt * J * 2;
  for (int i = threadIdx.x; i < J * 2; i += 256) pd[i] = ps[i];
}

extern "C" int pmce_assemble_windows_f32(const float* pose, const float* feat, const int* win, float* out_pose,
                                         float* out_feat, int W, int L, int J, hipStream_t stream) {
  PMCE_REQUIRE(pose && feat && win && out_pose && out_feat && W > 0 && L > 0 && J > 0, "assemble_windows: bad args");
  hipLaunchKernelGGL(assemble_windows_kernel, dim3(W * 16), dim3(256), 0, stream, pose, feat, win, out_pose, out_feat, W, L, J);
  return pmce_check_launch("assemble_windows");
}

// ------------------------------------------------------------------------------------------------------
// Detector output -> model input, per frame (data/PW3D/dataset.py:185-204,160-161,235-237): keep (x, y) of the J0 detected
// keypoints, append pelvis = (L_Hip + R_Hip)/2 and (unless only_pelvis) neck = (L_Shoulder + R_Shoulder)/2, then
// normalise to the screen: x' = x/w*2 - 1, y' = y/w*2 - h/w.  kp[L][J0][kp_stride] pixels, shape[L][2] = (height, width).
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prepare_pose2d_kernel(const float* __restrict__ kp, int kp_stride,
                                                             const int* __restrict__ shape, float* __restrict__ out, int L,
                                                             int J0, int n_extra, int lhip, int rhip, int lsho, int rsho) {
  const int J = J0 + n_extra;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= L * J) return;
  const int f = idx / J, j = idx % J;
  const float* k = kp + (long long)f * J0 * kp_stride;
  float x, y;
  if (j < J0) {
    x = k[j * kp_stride];
    y = k[j * kp_stride + 1];
  } else {
    const int a = (j == J0) ? lhip : lsho, b = (j == J0) ? rhip : rsho;
    x = (k[a * kp_stride] + k[b * kp_stride]) * 0.5f;
    y = (k[a * kp_stride + 1] + k[b * kp_stride + 1]) * 0.5f;
  }
  const float h = (float)shape[2 * f], w = (float)shape[2 * f + 1];
  out[(long long)idx * 2] = x / w * 2.0f - 1.0f;
  out[(long long)idx * 2 + 1] = y / w * 2.0f - h / w;
}

extern "C" int pmce_prepare_pose2d_f32(const float* kp, int kp_stride, const int* shape, float* out, int L, int J0, int n_extra,
                                       int lhip, int rhip, int lsho, int rsho, hipStream_t stream) {
  PMCE_REQUIRE(kp && shape && out && L > 0 && J0 > 0 && kp_stride >= 2, "prepare_pose2d: bad args");
  PMCE_REQUIRE(n_extra >= 0 && n_extra <= 2, "prepare_pose2d: n_extra must be 0 (none), 1 (pelvis) or 2 (pelvis + neck)");
  PMCE_REQUIRE(n_extra == 0 || (lhip >= 0 && lhip < J0 && rhip >= 0 && rhip < J0), "prepare_pose2d: hip index out of range");
  PMCE_REQUIRE(n_extra < 2 || (lsho >= 0 && lsho < J0 && rsho >= 0 && rsho < J0), "prepare_pose2d: shoulder index out of range");
  const long long n = (long long)L * (J0 + n_extra);
  hipLaunchKernelGGL(prepare_pose2d_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, kp, kp_stride, shape, out, L,
                     J0, n_extra, lhip, rhip, lsho, rsho);
  return pmce_check_launch("prepare_pose2d");
}
