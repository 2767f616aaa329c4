// Pose–mesh co-evolution decoder kernels (reference lib/models/CoevoDecoder.py), gfx950 only.
//
// All 64-channel token work is "token-local": a wavefront owns 32 tokens, the lane pair (l, l+32) holds one
// token's 64 channels in the slot layout of common.hpp, every Linear is a swapped-operand
// v_mfma_f32_32x32x2_f32 product  Y^T[n, tok] = W[n, k] * X^T[k, tok]  whose D registers are again the slot
// layout, so AdaLN -> Linear -> softmax -> Linear -> residual chains never leave registers and softmax /
// LayerNorm reductions are 16..32 in-lane ops plus one cross-half shuffle.
//
// Kernels (V = 431 vertices, D = 64 channels, J <= 32 joints):
//   vertex_init_gather   integer gather  vertxs = joints[:, vj_relation]                (CoevoDecoder.py:232)
//   ca_fold              per clip: joint embedding (jf, xk; :177-180,:184) + Wq folded into K and Wproj into V of the vertex<-joint
//                        cross-attention (+ the operands' f16 image for the split mode)
//   vertex_ca            fused AdaLN + vertex<-joint cross-attention + residual          (:83, :47-62, :23-29)   [fp32 pipe, stand-alone]
//   vertex_ca_mlp        the whole vertex-stream CrossAttentionBlock in one launch       (:82-87)                [both modes]
//   adaln_mlp            x + Mlp(AdaLN(x)) (+ Linear(64->3) + coordinate residual)       (:85-86,:104,:189)      [both modes]
//   adaln_qkv, vertex_sa qkv = Linear(64->192)(AdaLN(x)); flash-style 431x431 self-attention + proj + residual (:103,:118-131) [fp32 pipe]
//   vertex_sab           the same two in ONE launch, three-product f16 form                                      [split mode]
//   tokens_kv            k = Wk*AdaLN_k(xk)+bk, v = Wv*AdaLN_v(xv)+bv for the joint<-vertex direction
//   joint_stream         block-3 joint stream: joint<-vertex CA + FFN + SA + FFN + coords (:183,:187,:189)
#include "common.hpp"

#define NV 431   // down-sampled mesh vertices
#define ND 64    // joint_dim == vertx_dim (cfg.MODEL.joint_dim / vertx_dim)
#define NTILE 14 // ceil(431 / 32) wave tiles per clip
#define LDW64 68
#define LDW256 260

// ======================================================================================================
// vertex init gather — integer index path, bit-exact copy
// ======================================================================================================
__global__ __launch_bounds__(256) void vertex_init_gather_kernel(const float* __restrict__ joints,
                                                                 const int* __restrict__ vj, float* __restrict__ vt, int B,
                                                                 int J) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * NV * 3) return;
  const int l = idx % 3, v = (idx / 3) % NV, b = idx / (3 * NV);
  vt[idx] = joints[((long long)b * J + vj[v]) * 3 + l];
}

// ======================================================================================================
// small per-clip helpers for J-token work (VALU; J <= 32 tokens, tiny FLOPs)
// ======================================================================================================
#define JLD 65  // LDS row stride for [J][64] buffers

// ---- weight tiles for the per-clip kernels' Linear layers (joint_stream, ca_fold) --------------------------------------------------------------------------
// A 64 x 64 tile W[n0 .. n0+63][k0 .. k0+63] of a row-major [NOUT][KIN] nn.Linear weight travels global -> registers (coalesced:
// 16 lanes x 16 B cover a row's 256 bytes, a wave instruction four rows) -> LDS [64][JWL] (row stride 68 floats: 16-byte aligned
// rows, conflict-free ds_read_b128 with lane = row).  Until round 4 every lane walked its OWN weight row in global memory (64 cache
// lines per load instruction): 6,000 such instructions made 180 of the kernel's 233 us.
#define JWL 68
#define JS_THREADS 512
struct JTile {
  f32x4 r[2];  // 512 threads x 2 x 16 B = one 16 KB tile
};
__device__ __forceinline__ void jtile_fetch(JTile& t, const float* __restrict__ W, int ld, int n0, int k0, int tid) {
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int row = (tid >> 4) + 32 * q, c4 = tid & 15;
    t.r[q] = *reinterpret_cast<const f32x4*>(W + (long long)(n0 + row) * ld + k0 + 4 * c4);
  }
}
__device__ __forceinline__ void jtile_store(float* s_w, const JTile& t, int tid) {
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int row = (tid >> 4) + 32 * q, c4 = tid & 15;
    *reinterpret_cast<f32x4*>(s_w + row * JWL + 4 * c4) = t.r[q];
  }
}
// out[i][n] = act(b[n] + sum_k W[n][k] in[i][k]) for i < J: 512 threads = 64 output columns (lane) x 8 token groups (wave; tokens
// i = wave, wave + 8, ...: at most 4 for J <= 32).  The k order of every sum is 0, 1, 2, ... as in the reference's dot product.
// `pre` holds this layer's FIRST tile on entry (fetched by the previous phase, so its latency hid under that phase's arithmetic) and
// the NEXT layer's first tile (Wnext, ldnext; null = none) on return.
template <int KIN, int NOUT>
__device__ __forceinline__ void lin_tiled(const float* in, int ild, const float* __restrict__ W, const float* __restrict__ b, float* out,
                                          int old, int J, int tid, bool gelu, float* s_w, JTile& pre, const float* __restrict__ Wnext,
                                          int ldnext) {
  constexpr int NT = NOUT / 64, KT = KIN / 64;
  const int lane = tid & 63, tg = tid >> 6;
#pragma unroll 1
  for (int nt = 0; nt < NT; ++nt) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const int n = nt * 64 + lane;
    const float bn = b[n];  // requested here: in flight under the tile's arithmetic
#pragma unroll 1
    for (int kt = 0; kt < KT; ++kt) {
      __syncthreads();  // every wave is done with the previous tile in s_w (and, first time, `in` is complete)
      jtile_store(s_w, pre, tid);
      __syncthreads();
      // next tile of this layer, or the first tile of the next one: in flight under the arithmetic below
      if (kt + 1 < KT) jtile_fetch(pre, W, KIN, nt * 64, (kt + 1) * 64, tid);
      else if (nt + 1 < NT) jtile_fetch(pre, W, KIN, (nt + 1) * 64, 0, tid);
      else if (Wnext) jtile_fetch(pre, Wnext, ldnext, 0, 0, tid);
      const float* wrow = s_w + lane * JWL;
#pragma unroll 4
      for (int k4 = 0; k4 < 16; ++k4) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(wrow + 4 * k4);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int i = tg + 8 * t;
          if (i < J) {  // (wave-uniform)
            const float* x = in + i * ild + kt * 64 + 4 * k4;
            acc[t] = fmaf(wv.x, x[0], acc[t]);
            acc[t] = fmaf(wv.y, x[1], acc[t]);
            acc[t] = fmaf(wv.z, x[2], acc[t]);
            acc[t] = fmaf(wv.w, x[3], acc[t]);
          }
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int i = tg + 8 * t;
      if (i < J) {
        const float v = acc[t] + bn;
        out[i * old + n] = gelu ? gelu_erf(v) : v;
      }
    }
  }
}
// AdaLN over J tokens of 64 channels held in LDS: one wavefront per token (8 waves), lane = channel.
__device__ __forceinline__ void adaln_small8(const float* in, float* out, const float* __restrict__ gb, int J, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  const float gam = gb[lane], bet = gb[64 + lane];  // (one global load each, ahead of the token loop)
  for (int i = wave; i < J; i += 8) {
    const float x = in[i * JLD + lane];
    const float mean = wave_sum(x) * (1.0f / 64.0f);
    const float d = x - mean;
    const float var = wave_sum(d * d) * (1.0f / 63.0f);
    out[i * JLD + lane] = gam * d / (sqrtf(var) + 1e-6f) + bet;
  }
}

// ======================================================================================================
// ca_fold (joint_prep): per clip, the joint side of a CoevoBlock's vertex<-joint cross-attention.
//  (a) embed form (jt != null; CoevoDecoder.py:177-180,184): jf = joint_proj(jt) + joint_pos_embed ; xk = proj_j2v_dim(jf) + j2v_K_embed ;
//      xv = jf.  (Until round 5 a launch of its own, joint_embed; jf is written out only where the joint stream needs it.)
//  (b) fold: the query/output projections folded into the (tiny) key/value side.  With a = gamma_q*n + beta_q (n = normalised query
//      token), k = Wk*AdaLN_k(xk)+bk, v = Wv*AdaLN_v(xv)+bv, head h = channels [32h,32h+32):
//   score[h][i] = scale * (Wq a + bq)_h . k_i,h  =  n . Kf[h*32+i][:] + s0[h*32+i]
//   out         = sum_h P_h (v_h Wp_h^T) + bp     =  Vf[:, h*32+i] P[h*32+i] + bp
//   Kf[h*32+i][c] = scale*gamma_q[c]*M ; s0 = scale*(sum_c beta_q[c]*M + bq_h.k_i,h) ; M = sum_d Wq[32h+d][c]*k_i[32h+d]
//   Vf[c][h*32+i] = sum_d v_i[32h+d]*Wp[c][32h+d]
// Rows/columns i >= J are zero (masked to -inf in vertex_ca).  Algebraically identical to
// CrossAttention.forward (CoevoDecoder.py:47-62) after AdaLN (:83); only the summation order differs.
// gbq/gbk/gbv: this clip's [gamma|beta] (128 floats) of normq/normk/normv.
//  (c) image (img != null, J <= 23; round 5): the same operands as the LDS image vertex_ca_mlp's f16 form copies - per clip
//      [s0 (64) | 2^-sK 2^-10, 2^-sV 2^-10, 0, 0 | Kf rows (h, i < J) as (hi | lo) f16 A fragments of Kf * 2^sK, stage_weight_split's row
//      format | Vf rows c as (hi | lo) fragments of Vf * 2^sV over the keys: per head k-step 0 (keys 0 .. 15: 2 lane halves x (8 hi | 8 lo))
//      then k-step 1 (keys 16 .. 23: 2 lane halves x (4 hi | 4 lo))], one power of two per operand and clip (max |.| -> [2^14, 2^15)).
// ======================================================================================================
#define CAM_VLD 52  // Vf compact row stride: 2 heads x 24 keys + 4 (conflict-free ds_read_b128 across 32 rows)
#define CA_IMG_HDR 68
#define CA_IMG_FLOATS(J) (CA_IMG_HDR + 2 * (J) * LDW64 + 64 * CAM_VLD)
#define CA_IMG_STRIDE 6528  // floats per clip: CA_IMG_FLOATS(23) = 6524 rounded up to 256 bytes
struct CaFoldArgs {
  const float *xk, *xv;                                       // given token features [B,J,64] (xv = the joint features) ...
  const float *jt, *Wj, *bj, *jpos, *Wj2v, *bj2v, *j2vK;      // ... or (xk == null) the embed form: joints [B,J,3] + the embedding weights
  float* jf_out;                                              // embed form: jf [B,J,64] written here when non-null
  const float* GB;
  int gb_stride, iq, ik, iv;
  const float *Wq, *bq, *Wk, *bk, *Wv, *bv, *Wp;
  float *Kf, *s0, *Vf;                                        // fp32 folded operands (any may be null)
  float* img;                                                 // f16 image [B][CA_IMG_STRIDE] or null
  int J;
};
__global__ __launch_bounds__(JS_THREADS) void ca_fold_kernel(CaFoldArgs a) {
  __shared__ float s_a[32 * JLD];
  __shared__ float s_b[32 * JLD];
  __shared__ float s_k[32 * JLD];
  __shared__ float s_v[32 * JLD];
  __shared__ __attribute__((aligned(16))) float s_w[64 * JWL];    // Wj2v, Wk, Wv, then Wq (64 x 64 tiles through lin_tiled's ring of one)
  __shared__ __attribute__((aligned(16))) float s_wp[64 * JWL];   // Wproj
  __shared__ __attribute__((aligned(16))) float s_img[CA_IMG_STRIDE];
  __shared__ float s_red[16];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int J = a.J;
  const float* gb = a.GB + (long long)b * a.gb_stride;
  // all 64 x 64 weights travel as coalesced tiles (until round 4 each lane walked its own weight row in global memory)
  JTile pre, prep;
  jtile_fetch(pre, a.xk ? a.Wk : a.Wj2v, 64, 0, 0, tid);
  jtile_fetch(prep, a.Wp, 64, 0, 0, tid);
  const float gq = gb[a.iq * 128 + lane], bqv = gb[a.iq * 128 + 64 + lane];
  if (a.img)
    for (int i = tid; i < CA_IMG_STRIDE; i += JS_THREADS) s_img[i] = 0.f;  // pads and dead keys: true zeros
  if (a.xk) {
    for (int idx = tid; idx < J * 64; idx += JS_THREADS) {
      const int i = idx >> 6, c = idx & 63;
      s_a[i * JLD + c] = a.xk[((long long)b * J + i) * 64 + c];
      s_b[i * JLD + c] = a.xv[((long long)b * J + i) * 64 + c];
    }
    jtile_store(s_wp, prep, tid);
    __syncthreads();
  } else {  // joint_embed: jf -> s_b ; proj_j2v_dim(jf) + j2v_K_embed -> s_a
    // 512 threads = 8 tokens x 64 channels per pass, at most 4 passes (J <= 32); a thread's channel is the same in every pass.  Every
    // load of the phase is issued before its first use (as plain loops each pass waited for its own dependent loads).
    const int c = tid & 63, i0 = tid >> 6;
    const float w0 = a.Wj[c * 3], w1 = a.Wj[c * 3 + 1], w2 = a.Wj[c * 3 + 2], bjc = a.bj[c];
    float px[4], py[4], pz[4], pe[4], ke[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + 8 * u;
      if (i < J) {
        const float* p = a.jt + ((long long)b * J + i) * 3;
        px[u] = p[0]; py[u] = p[1]; pz[u] = p[2];
        pe[u] = a.jpos[i * 64 + c];
        ke[u] = a.j2vK[i * 64 + c];   // (needed after the product below: in flight meanwhile)
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + 8 * u;
      if (i < J) {
        const float v = ((w0 * px[u] + w1 * py[u] + w2 * pz[u]) + bjc) + pe[u];
        s_b[i * JLD + c] = v;
        if (a.jf_out) a.jf_out[((long long)b * J + i) * 64 + c] = v;
      }
    }
    jtile_store(s_wp, prep, tid);
    lin_tiled<64, 64>(s_b, JLD, a.Wj2v, a.bj2v, s_k, JLD, J, tid, false, s_w, pre, a.Wk, 64);  // (its first barrier publishes s_b and s_wp)
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + 8 * u;
      if (i < J) s_a[i * JLD + c] = s_k[i * JLD + c] + ke[u];
    }
    __syncthreads();
  }
  adaln_small8(s_a, s_k, gb + a.ik * 128, J, tid);  // s_k = AdaLN_k(xk)
  adaln_small8(s_b, s_v, gb + a.iv * 128, J, tid);  // s_v = AdaLN_v(xv)
  lin_tiled<64, 64>(s_k, JLD, a.Wk, a.bk, s_a, JLD, J, tid, false, s_w, pre, a.Wv, 64);  // s_a = k
  lin_tiled<64, 64>(s_v, JLD, a.Wv, a.bv, s_b, JLD, J, tid, false, s_w, pre, a.Wq, 64);  // s_b = v
  __syncthreads();
  jtile_store(s_w, pre, tid);  // Wq
  __syncthreads();
  // 32^-0.5 (vertx heads = 2, head_dim 32; CoevoDecoder.py:140,37-38) times log2(e): vertex_ca's softmax runs on the
  // hardware 2^x, so the folded scores are produced directly in log2 units
  const float scale = 0.17677669529663688110f * 1.44269504088896340736f;
  float* Kfb = a.Kf ? a.Kf + (long long)b * 64 * 64 : nullptr;
  float* Vfb = a.Vf ? a.Vf + (long long)b * 64 * 64 : nullptr;
  float* s0b = a.s0 ? a.s0 + (long long)b * 64 : nullptr;
  // one wavefront per (h,i) row, lane = channel c: rows wave, wave + 8, ... (8 per wave)
  float kfr[8], vfr[8];
  float mk = 0.f, mv = 0.f;
#pragma unroll
  for (int q8 = 0; q8 < 8; ++q8) {
    const int row = wave + 8 * q8;
    const int h = row >> 5, i = row & 31;
    float kf = 0.f, vf = 0.f, sc = 0.f;
    if (i < J) {
      float M = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const f32x4 wp4 = *reinterpret_cast<const f32x4*>(s_wp + lane * JWL + 32 * h + 4 * q);   // Wp[c = lane][32h + 4q ..]
        const float* kd = s_a + i * JLD + 32 * h + 4 * q;
        const float* vd = s_b + i * JLD + 32 * h + 4 * q;
        const float* wq = s_w + (32 * h + 4 * q) * JWL + lane;                                   // Wq[32h + d][c = lane]
        M += wq[0] * kd[0];
        vf += vd[0] * wp4.x;
        M += wq[JWL] * kd[1];
        vf += vd[1] * wp4.y;
        M += wq[2 * JWL] * kd[2];
        vf += vd[2] * wp4.z;
        M += wq[3 * JWL] * kd[3];
        vf += vd[3] * wp4.w;
      }
      kf = scale * gq * M;
      float t = bqv * M;
      if (lane < 32) t += a.bq[32 * h + lane] * s_a[i * JLD + 32 * h + lane];
      sc = scale * wave_sum(t);
    }
    if (Kfb) Kfb[row * 64 + lane] = kf;
    if (Vfb) Vfb[lane * 64 + row] = vf;
    if (s0b && lane == 0) s0b[row] = sc;
    if (a.img && lane == 0) s_img[row] = sc;
    kfr[q8] = kf;
    vfr[q8] = vf;
    mk = fmaxf(mk, fabsf(kf));
    mv = fmaxf(mv, fabsf(vf));
  }
  if (!a.img) return;  // (uniform over the grid)
  mk = wave_max(mk);
  mv = wave_max(mv);
  if (lane == 0) {
    s_red[wave] = mk;
    s_red[8 + wave] = mv;
  }
  __syncthreads();
  float MK = 0.f, MV = 0.f;
#pragma unroll
  for (int w = 0; w < 8; ++w) {
    MK = fmaxf(MK, s_red[w]);
    MV = fmaxf(MV, s_red[8 + w]);
  }
  int ek = 0, ev = 0;
  if (MK > 0.f) frexpf(MK, &ek);
  if (MV > 0.f) frexpf(MV, &ev);
  const float upK = MK > 0.f ? ldexpf(1.f, 15 - ek) : 1.f, upV = MV > 0.f ? ldexpf(1.f, 15 - ev) : 1.f;
  if (tid == 0) {  // what the kernel multiplies its accumulators by: the operand's 2^-s and the 2^-10 of its register-side operand
    s_img[64] = (MK > 0.f ? ldexpf(1.f, ek - 15) : 1.f) * 0.0009765625f;
    s_img[65] = (MV > 0.f ? ldexpf(1.f, ev - 15) : 1.f) * 0.0009765625f;
  }
  _Float16* im = reinterpret_cast<_Float16*>(s_img);
#pragma unroll
  for (int q8 = 0; q8 < 8; ++q8) {
    const int row = wave + 8 * q8;
    const int h = row >> 5, i = row & 31;
    if (i < J) {  // Kf row (h, i), channel c = lane -> k-step c / 16, lane half (c / 4) & 1, element (c & 3) + 4 ((c / 8) & 1)
      const float v = pinned(kfr[q8] * upK);  // ONE fp32 value for both planes (common.hpp)
      const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
      const int c = lane, ks = c >> 4, hbb = (c >> 2) & 1, e = (c & 3) + 4 * ((c >> 3) & 1);
      const int idx = (CA_IMG_HDR + (h * J + i) * LDW64) * 2 + ((ks * 2 + hbb) * 2) * 8 + e;
      im[idx] = hi;
      im[idx + 8] = lo;
    }
    if (i < 24) {  // Vf[c = lane][key (h, i)] (zero for J <= i < 24)
      const float v = pinned(vfr[q8] * upV);
      const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
      const int base = (CA_IMG_HDR + 2 * J * LDW64 + lane * CAM_VLD) * 2 + h * 48;
      if (i < 16) {
        const int idx = base + ((i >> 2) & 1) * 16 + (i & 3) + 4 * (i >> 3);
        im[idx] = hi;
        im[idx + 8] = lo;
      } else {
        const int jj = i - 16;
        const int idx = base + 32 + (jj >> 2) * 8 + (jj & 3);
        im[idx] = hi;
        im[idx + 4] = lo;
      }
    }
  }
  __syncthreads();
  float* dst = a.img + (long long)b * CA_IMG_STRIDE;
  for (int i4 = tid; i4 < CA_IMG_FLOATS(J) / 4; i4 += JS_THREADS) reinterpret_cast<f32x4*>(dst)[i4] = reinterpret_cast<const f32x4*>(s_img)[i4];
}

// ======================================================================================================
// vertex_ca — the north-star kernel: fused AdaLN + vertex<-joint cross-attention + residual.
//   out[b][v][:] = xq + proj(softmax((Wq AdaLN_q(xq))(Wk AdaLN_k(xk))^T / sqrt(32)) (Wv AdaLN_v(xv)))
// with the key/value side pre-folded by ca_fold.  xq is either read ([B,431,64]) or, when xq == nullptr,
// formed on the fly from the 3-D vertex coordinates: xq = Wv3*vt + Eq[v]  (Eq = vertx_proj.bias +
// vertx_pos_embed + v_Q_embed, CoevoDecoder.py:177-180,184).
// grid (2, B), 448 threads: 7 waves stage the clip's folded operands into LDS and each own 32 vertices.
// Per 32 vertices: 8 x 16-B loads/lane, 128 MFMAs (64 scores + 64 output), masked softmax over <= 32 keys as
// 16 in-lane values + one shuffle, 8 x 16-B stores/lane.
// ======================================================================================================
__global__ __launch_bounds__(448, 4) void vertex_ca_kernel(const float* __restrict__ xq, const float* __restrict__ vt,
                                                        const float* __restrict__ Wv3, const float* __restrict__ Eq,
                                                        const float* __restrict__ Kf, const float* __restrict__ s0,
                                                        const float* __restrict__ Vf, const float* __restrict__ bp,
                                                        float* __restrict__ out, int J) {
  __shared__ __attribute__((aligned(16))) float sK[64 * LDW64];
  __shared__ __attribute__((aligned(16))) float sV[64 * LDW64];
  __shared__ float sS0[64];
  const int b = blockIdx.y, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int n0 = lane & 31, hb = lane >> 5;
  const int tile = blockIdx.x * 7 + wave;
  const int v = tile * 32 + n0;
  const bool valid = v < NV;
  const int vc = valid ? v : NV - 1;

  // this wave's 32 query tokens first: their HBM latency hides under the staging of the clip's folded operands
  float x[32];
  if (xq) {
    load_slots(xq + ((long long)b * NV + vc) * 64, x, hb);
  } else {
    const float* p = vt + ((long long)b * NV + vc) * 3;
    const float p0 = p[0], p1 = p[1], p2 = p[2];
    load_slots(Eq + (long long)vc * 64, x, hb);
#pragma unroll
    for (int s = 0; s < 32; ++s) {
      const int c = slot_channel(s, hb);
      x[s] = (Wv3[c * 3] * p0 + Wv3[c * 3 + 1] * p1 + Wv3[c * 3 + 2] * p2) + x[s];
    }
  }
  stage_weight<64>(sK, Kf + (long long)b * 4096, 64, tid, 448);
  stage_weight<64>(sV, Vf + (long long)b * 4096, 64, tid, 448);
  if (tid < 64) sS0[tid] = s0[(long long)b * 64 + tid];
  // normalise (gamma/beta are folded into Kf/s0)
  float n[32];
  {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += x[i];
    const float mean = pair_sum(s) * (1.0f / 64.0f);
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const float d = x[i] - mean;
      ss += d * d;
    }
    const float inv = 1.0f / (sqrtf(pair_sum(ss) * (1.0f / 63.0f)) + 1e-6f);
#pragma unroll
    for (int i = 0; i < 32; ++i) n[i] = (x[i] - mean) * inv;
  }
  __syncthreads();
  // scores^T[h*32+i, tok]
  f32x16 sc[2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[h][r] = sS0[h * 32 + (r & 3) + 8 * (r >> 2) + 4 * hb];
  tl_gemm<8, 2, LDW64>(sK, n, sc, n0, hb);
  // masked softmax over the joints of each head: 16 in-lane rows + the partner lane's 16
  float p[32];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * hb;
      const float sv = (i < J) ? sc[h][r] : -INFINITY;
      sc[h][r] = sv;
      m = fmaxf(m, sv);
    }
    m = pair_max(m);
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = __builtin_amdgcn_exp2f(sc[h][r] - m);  // scores are in log2 units (ca_fold)
      p[16 * h + r] = e;
      sum += e;
    }
    const float inv = 1.0f / pair_sum(sum);
#pragma unroll
    for (int r = 0; r < 16; ++r) p[16 * h + r] *= inv;
  }
  // out^T[c, tok] = Vf[c, :] P + bp[c] + xq.  k-slot group qq of head h covers joints 8qq .. 8qq+7: groups whose
  // joints are all >= J carry P = 0 and are skipped (J = 17: 3 of 4 groups per head remain).
  f32x16 o[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[nt][r] = bp[nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hb];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    if (8 * (q & 3) < J) {  // wave-uniform
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(sV + (nt * 32 + n0) * LDW64 + 8 * q + 4 * hb);
        o[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, p[4 * q + 0], o[nt], 0, 0, 0);
        o[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, p[4 * q + 1], o[nt], 0, 0, 0);
        o[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, p[4 * q + 2], o[nt], 0, 0, 0);
        o[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, p[4 * q + 3], o[nt], 0, 0, 0);
      }
    }
  }
  if (valid) {
    float y[32];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) y[16 * nt + r] = x[16 * nt + r] + o[nt][r];
    store_slots(out + ((long long)b * NV + v) * 64, y, hb);
  }
}


// ======================================================================================================
// The 64 -> 256 -> 64 AdaLN-FFN of a 32-token wave tile (shared by adaln_mlp and vertex_ca_mlp), two arithmetic forms.
//  F16 = false: fp32 matrix pipe (tl_gemm), weights staged as fp32.
//  F16 = true : the three-product f16 form of gemm_split_f16.hip.  A token's activation in slot layout is, per 16-channel k-step
//    s, the registers [8s, 8s+8) of a lane - exactly one f16x8 B fragment once split into (hi, lo*2^11) - and the weights are
//    staged into LDS as the matching A fragments: row r holds, per k-step s and lane half hb, 8 f16 hi then 8 f16 lo of
//    W[r][c(s,hb,e)] * 2^sw, c(s,hb,e) = 16s + 4hb + e (e < 4), 16s + 8 + 4hb + e - 4 (e >= 4): the bytes and the row stride of the
//    fp32 staging.  acc_main += whi*xhi + wlo*xhi, acc_corr += whi*xlo (lo carries 2^11), result = (acc_main + acc_corr*2^-11)*2^-sw:
//    12 + 12 matrix instructions of 32 cycles per 32 hidden units instead of 32 + 32 of 64 cycles.
// ======================================================================================================
typedef _Float16 tl_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 tl_f16x4 __attribute__((ext_vector_type(4)));

// fp32 [rows][K] global -> LDS [rows][K+4 floats] of f16 fragments (see above), scaled by `up` (a power of two)
template <int K>
__device__ __forceinline__ void stage_weight_split(float* __restrict__ dst, const float* __restrict__ src, int rows, float up,
                                                   int tid, int nthreads) {
  constexpr int C4 = K / 4;
  for (int i = tid; i < rows * C4; i += nthreads) {
    const int r = i / C4, c = i % C4;  // channels 4c .. 4c+3 of row r
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + (size_t)r * K + 4 * c) * up;
    tl_f16x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) hi[e] = (_Float16)v[e];
#pragma unroll
    for (int e = 0; e < 4; ++e) lo[e] = (_Float16)(v[e] - (float)hi[e]);
    const int ks = c >> 2, g = c & 3;  // k-step, group of 4 inside it: g = 0: hb 0 e 0..3, 1: hb 1 e 0..3, 2: hb 0 e 4..7, 3: hb 1 e 4..7
    _Float16* chunk = reinterpret_cast<_Float16*>(dst + r * (K + 4) + ((ks * 2 + (g & 1)) * 2) * 4) + (g >> 1) * 4;
    *reinterpret_cast<tl_f16x4*>(chunk) = hi;
    *reinterpret_cast<tl_f16x4*>(chunk + 8) = lo;
  }
}
// max |w| of a weight over the workgroup -> (2^s, 2^-s) with max|w| * 2^s in [2^14, 2^15); red: 16 floats of LDS scratch
__device__ __forceinline__ void weight_scale(const float* __restrict__ src, int n, float* red, int tid, int nthreads, float& up,
                                            float& down) {
  float m = 0.f;
  for (int i = tid; i < n / 4; i += nthreads) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + 4 * i);
    m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(m, fmaxf(fabsf(v.z), fabsf(v.w))));
  }
  m = wave_max(m);
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  float mx = 0.f;
  for (int w = 0; w < (nthreads + 63) / 64; ++w) mx = fmaxf(mx, red[w]);
  __syncthreads();
  int e = 0;
  if (mx > 0.f) frexpf(mx, &e);
  up = mx > 0.f ? ldexpf(1.f, 15 - e) : 1.f;
  down = mx > 0.f ? ldexpf(1.f, e - 15) : 1.f;
}
// slots [8s, 8s+8) of an activation -> the (hi, lo*2^11) B fragments of k-step s
__device__ __forceinline__ void split_slots8(const float* x_, tl_f16x8& hi, tl_f16x8& lo) {
  float x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) x[e] = pinned(x_[e]);  // one fp32 value for both planes (common.hpp)
#pragma unroll
  for (int e = 0; e < 8; ++e) hi[e] = (_Float16)x[e];
#pragma unroll
  for (int e = 0; e < 8; ++e) lo[e] = (_Float16)((x[e] - (float)hi[e]) * 2048.0f);
}

// the same with lo at its true magnitude (for operands that carry their own power of two).  lo = rne16(x - hi) is one mixed-precision
// multiply-add per element (v_fma_mix{lo,hi}_f16: fp32 x * 1.0 - f16 hi, rounded to f16) instead of convert back / subtract / convert.
typedef _Float16 tl_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair_plain(float x0, float x1, tl_f16x2& hi, tl_f16x2& lo) {
  hi[0] = (_Float16)x0;
  hi[1] = (_Float16)x1;
  const unsigned H = __builtin_bit_cast(unsigned, hi);
  unsigned L;
  asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(L) : "v"(x0), "v"(H));
  asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(L) : "v"(x1), "v"(H));
  lo = __builtin_bit_cast(tl_f16x2, L);
}
__device__ __forceinline__ void split_slots8_plain(const float* x, tl_f16x8& hi, tl_f16x8& lo) {
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    tl_f16x2 h, l;
    split_pair_plain(x[e], x[e + 1], h, l);
    hi[e] = h[0], hi[e + 1] = h[1];
    lo[e] = l[0], lo[e + 1] = l[1];
  }
}
__device__ __forceinline__ void split4_plain(float x0, float x1, float x2, float x3, tl_f16x4& hi, tl_f16x4& lo) {
  tl_f16x2 h0, l0, h1, l1;
  split_pair_plain(x0, x1, h0, l0);
  split_pair_plain(x2, x3, h1, l1);
  hi = tl_f16x4{h0[0], h0[1], h1[0], h1[1]};
  lo = tl_f16x4{l0[0], l0[1], l1[0], l1[1]};
}

// a: AdaLN output of the tile (32 slots); x: the residual input; returns y = x + fc2(gelu(fc1(a))) in acc2 (D layout = slot layout)
template <bool F16>
__device__ __forceinline__ void ffn_slots(const float* sW1, const float* sW2, const float* sB1, const float* sB2, const float* sSc,
                                          const float* a, const float* x, f32x16 (&acc2)[2], int n0, int hb) {
  if constexpr (!F16) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[nt][r] = sB2[nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hb] + x[16 * nt + r];
#pragma unroll 1
    for (int ht = 0; ht < 8; ++ht) {
      f32x16 acc1;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[r] = sB1[ht * 32 + (r & 3) + 8 * (r >> 2) + 4 * hb];
      tl_gemm<8, 1, LDW64>(sW1 + ht * 32 * LDW64, a, &acc1, n0, hb);
      float hreg[16];
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 g = gelu_erf2(f32x2{acc1[r], acc1[r + 1]});
        hreg[r] = g.x;
        hreg[r + 1] = g.y;
      }
      tl_gemm<4, 2, LDW256>(sW2 + ht * 32, hreg, acc2, n0, hb);
    }
  } else {
    const float down1 = sSc[1], down2 = sSc[3];
    tl_f16x8 ahi[4], alo[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) split_slots8(a + 8 * s, ahi[s], alo[s]);
    f32x16 m2[2], c2[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) m2[nt][r] = c2[nt][r] = 0.f;
#pragma unroll 1
    for (int ht = 0; ht < 8; ++ht) {
      f32x16 m1, c1;
#pragma unroll
      for (int r = 0; r < 16; ++r) m1[r] = c1[r] = 0.f;
      const float* w1 = sW1 + (ht * 32 + n0) * LDW64 + hb * 8;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const tl_f16x8 whi = *reinterpret_cast<const tl_f16x8*>(w1 + s * 16), wlo = *reinterpret_cast<const tl_f16x8*>(w1 + s * 16 + 4);
        m1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, ahi[s], m1, 0, 0, 0);
        m1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo, ahi[s], m1, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, alo[s], c1, 0, 0, 0);
      }
      float hreg[16];
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const int u = ht * 32 + (r & 3) + 8 * (r >> 2) + 4 * hb;  // r and r+1: consecutive hidden units (r even)
        const f32x2 pre = {fmaf(fmaf(c1[r], 0.00048828125f, m1[r]), down1, sB1[u]),
                           fmaf(fmaf(c1[r + 1], 0.00048828125f, m1[r + 1]), down1, sB1[u + 1])};
        const f32x2 g = gelu_erf2(pre);
        hreg[r] = g.x;
        hreg[r + 1] = g.y;
      }
      tl_f16x8 hhi[2], hlo[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) split_slots8(hreg + 8 * s, hhi[s], hlo[s]);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const float* w2 = sW2 + (nt * 32 + n0) * LDW256 + ht * 32 + hb * 8;  // k-steps 2 ht, 2 ht + 1 of the 256-wide row
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const tl_f16x8 whi = *reinterpret_cast<const tl_f16x8*>(w2 + s * 16), wlo = *reinterpret_cast<const tl_f16x8*>(w2 + s * 16 + 4);
          m2[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, hhi[s], m2[nt], 0, 0, 0);
          m2[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo, hhi[s], m2[nt], 0, 0, 0);
          c2[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, hlo[s], c2[nt], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        acc2[nt][r] = fmaf(fmaf(c2[nt][r], 0.00048828125f, m2[nt][r]), down2, sB2[nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hb]) + x[16 * nt + r];
  }
}
// stage the FFN's two weights (and, in the f16 form, their scales into sSc[4] = {2^s1, 2^-s1, 2^s2, 2^-s2})
template <bool F16>
__device__ __forceinline__ void stage_ffn(float* sW1, float* sW2, float* sSc, const float* W1, const float* W2, int tid, int nthreads) {
  if constexpr (F16) {
    float up1, down1, up2, down2;
    weight_scale(W1, 256 * 64, sSc + 4, tid, nthreads, up1, down1);
    weight_scale(W2, 64 * 256, sSc + 4, tid, nthreads, up2, down2);
    if (tid == 0) {
      sSc[0] = up1; sSc[1] = down1; sSc[2] = up2; sSc[3] = down2;
    }
    stage_weight_split<64>(sW1, W1, 256, up1, tid, nthreads);
    stage_weight_split<256>(sW2, W2, 64, up2, tid, nthreads);
  } else {
    stage_weight<64>(sW1, W1, 256, tid, nthreads);
    stage_weight<256>(sW2, W2, 64, tid, nthreads);
  }
}

// The f16 form's LDS image of an FFN, made ONCE per model (pmce_ffn_pack_f16: [sW1 | sW2] as stage_ffn<true> writes them, then the four
// scales): staging a workgroup's 137 KB is then a plain copy instead of three passes over the fp32 weights (two for the scales, one
// converting) - a launch of the two FFN-carrying kernels is 1.75 wave tiles per wave slot, so that staging was a third of its time.
#define FFN_IMG_FLOATS (256 * LDW64 + 64 * LDW256 + 32)
template <bool F16>
__device__ __forceinline__ void stage_ffn_any(float* sW1, float* sW2, float* sSc, const float* W1, const float* W2,
                                              const float* __restrict__ img, int tid, int nthreads) {
  if constexpr (F16) {
    if (img) {  // 133 KiB, a plain copy: by LDS-DMA, every piece in flight at once (the caller's barrier is preceded by lds_dma_wait)
      static_assert((256 * LDW64 + 64 * LDW256) * 4 % 1024 == 0, "the FFN image is a whole number of KiB");
      lds_dma_copy(sW1, img, (256 * LDW64 + 64 * LDW256) * 4 / 1024, tid >> 6, nthreads >> 6, tid & 63);
      if (tid < 4) sSc[tid] = img[256 * LDW64 + 64 * LDW256 + tid];
      return;
    }
  }
  stage_ffn<F16>(sW1, sW2, sSc, W1, W2, tid, nthreads);
}
__global__ __launch_bounds__(512) void ffn_pack_kernel(const float* __restrict__ W1, const float* __restrict__ W2, float* __restrict__ img) {
  stage_ffn<true>(img, img + 256 * LDW64, img + 256 * LDW64 + 64 * LDW256, W1, W2, threadIdx.x, 512);
}
extern "C" int pmce_ffn_pack_f16(const float* W1, const float* W2, float* img, hipStream_t stream) {
  PMCE_REQUIRE(W1 && W2 && img && (reinterpret_cast<uintptr_t>(img) & 15) == 0, "ffn_pack: null or unaligned pointer");
  hipLaunchKernelGGL(ffn_pack_kernel, dim3(1), dim3(512), 0, stream, W1, W2, img);
  return pmce_check_launch("ffn_pack_f16");
}
extern "C" int pmce_ffn_image_floats(void) { return FFN_IMG_FLOATS; }

// ======================================================================================================
// adaln_mlp:  y = x + fc2(gelu(fc1(AdaLN(x))))   (hidden 256), optionally followed by the coordinate head
//   vt_out = Wc*y + bc + vt_in  (proj_vertx_feat2coor + residual, CoevoDecoder.py:189).
// Persistent workgroups (4 waves); fc1/fc2 weights live in LDS (137 KB) for the whole kernel; each wave walks
// over 32-token tiles; the 256-wide hidden activation exists only as 16 registers at a time.
// ======================================================================================================
template <bool F16, int NW = 8>
__global__ __launch_bounds__(64 * NW) void adaln_mlp_kernel(const float* __restrict__ xin, const float* __restrict__ GB,
                                                        int gb_stride, int inst, const float* __restrict__ W1,
                                                        const float* __restrict__ b1, const float* __restrict__ W2,
                                                        const float* __restrict__ b2, float* __restrict__ yout,
                                                        const float* __restrict__ Wc, const float* __restrict__ bc,
                                                        const float* __restrict__ vt_in, float* __restrict__ vt_out, int B,
                                                        const float* __restrict__ ffn_img) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sW1 = smem;                    // [256][68]
  float* sW2 = sW1 + 256 * LDW64;       // [64][260]
  float* sB1 = sW2 + 64 * LDW256;       // [256]
  float* sB2 = sB1 + 256;               // [64]
  float* sSc = sB2 + 64;                // [4 + 16] scales of the f16 form + reduction scratch (+ 12 pad)
  float* sWc = sSc + 32;                // [3][64] coordinate head + [3] its bias (fetched once, not per tile after the FFN)
  const int tid = threadIdx.x;
  stage_ffn_any<F16>(sW1, sW2, sSc, W1, W2, ffn_img, tid, 64 * NW);
  if (tid < 256) sB1[tid] = b1[tid];
  if (tid < 64) sB2[tid] = b2[tid];
  if (vt_out) {
    if (tid < 192) sWc[tid] = Wc[tid];
    if (tid < 3) sWc[192 + tid] = bc[tid];
  }
  lds_dma_wait();
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  const int n0 = lane & 31, hb = lane >> 5;
  const int ntiles = B * NTILE;
  for (int wt = blockIdx.x * NW + wave; wt < ntiles; wt += gridDim.x * NW) {
    const int b = wt / NTILE, tile = wt % NTILE;
    const int v = tile * 32 + n0;
    const bool valid = v < NV;
    const long long tok = (long long)b * NV + (valid ? v : NV - 1);
    float x[32], a[32];
    load_slots(xin + tok * 64, x, hb);
    adaln_slots(x, a, GB + (long long)b * gb_stride + inst * 128, hb);
    f32x16 acc2[2];
    ffn_slots<F16>(sW1, sW2, sB1, sB2, sSc, a, x, acc2, n0, hb);
    if (yout && valid) {
      float y[32];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) y[16 * nt + r] = acc2[nt][r];
      store_slots(yout + tok * 64, y, hb);
    }
    if (vt_out) {
      float d0 = 0.f, d1 = 0.f, d2 = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(sWc + 8 * q + 4 * hb);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(sWc + 64 + 8 * q + 4 * hb);
        const f32x4 w2 = *reinterpret_cast<const f32x4*>(sWc + 128 + 8 * q + 4 * hb);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int s = 4 * q + i;
          const float yv = acc2[s >> 4][s & 15];
          d0 += w0[i] * yv;
          d1 += w1[i] * yv;
          d2 += w2[i] * yv;
        }
      }
      d0 = pair_sum(d0);
      d1 = pair_sum(d1);
      d2 = pair_sum(d2);
      if (valid && hb == 0) {
        const float* pi = vt_in + tok * 3;
        float* po = vt_out + tok * 3;
        po[0] = (d0 + sWc[192]) + pi[0];
        po[1] = (d1 + sWc[193]) + pi[1];
        po[2] = (d2 + sWc[194]) + pi[2];
      }
    }
  }
}

// ======================================================================================================
// vertex_ca_mlp - the whole CrossAttentionBlock of the vertex stream in one launch (CoevoDecoder.py:82-87):
//   F1 = xq + CA(AdaLN_q(xq), AdaLN_k(xk), AdaLN_v(xv))   (vertex_ca above, key/value side pre-folded by ca_fold)
//   F2 = F1 + Mlp(AdaLN_2(F1))                             (adaln_mlp above)
// F1 never leaves the registers of the wave that made it.  Persistent workgroups of 7 waves keep fc1/fc2 in LDS (137 KB) for
// the whole launch, which leaves 26 KB: enough for ONE clip's folded operands in compact form - of Kf only the J live
// rows per head, of Vf only the key groups below J (3 of 4 per head up to J = 24) - so a work item is half a clip
// (7 wave tiles, one per wave, like vertex_ca's grid) and the operands are re-staged between items.  J <= 23; the
// launcher falls back to the two kernels above beyond that.  Same arithmetic in the same order as vertex_ca + adaln_mlp.
// ======================================================================================================
template <bool F16, int NW = 7>
__global__ __launch_bounds__(64 * NW) void vertex_ca_mlp_kernel(const float* __restrict__ xq, const float* __restrict__ vt,
                                                            const float* __restrict__ Wv3, const float* __restrict__ Eq,
                                                            const float* __restrict__ Kf, const float* __restrict__ s0,
                                                            const float* __restrict__ Vf, const float* __restrict__ bp,
                                                            const float* __restrict__ GB, int gb_stride, int inst,
                                                            const float* __restrict__ W1, const float* __restrict__ b1,
                                                            const float* __restrict__ W2, const float* __restrict__ b2,
                                                            float* __restrict__ yout, int B, int J, const float* __restrict__ ffn_img,
                                                            const float* __restrict__ ca_img) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sW1 = smem;                    // [256][68]
  float* sW2 = sW1 + 256 * LDW64;       // [64][260]
  float* sB1 = sW2 + 64 * LDW256;       // [256]
  float* sB2 = sB1 + 256;               // [64]
  float* sSc = sB2 + 64;                // [4 + 16] scales of the f16 FFN form + reduction scratch (+ 12 pad)
  // one clip's folded cross-attention operands, in the order of ca_fold's image (F16: a plain copy of it, f16 planes; fp32 form: the
  // compact fp32 rows): [s0 (64) | 4 scales | Kf rows (h, i < J) | Vf rows c over the keys (h, i < 24)]
  float* sS0 = sSc + 32;                // [2][32]
  float* sK = sS0 + CA_IMG_HDR;         // [2*J][68]       Kf[h*J + i][c], i < J
  float* sV = sK + 2 * J * LDW64;       // [64][CAM_VLD]   Vf[c][h*24 + i], i < 24
  const int tid = threadIdx.x;
  stage_ffn_any<F16>(sW1, sW2, sSc, W1, W2, ffn_img, tid, 64 * NW);
  if (tid < 256) sB1[tid] = b1[tid];
  if (tid < 64) sB2[tid] = b2[tid];
  const int lane = tid & 63, wave = tid >> 6;
  const int n0 = lane & 31, hb = lane >> 5;
  const int krow = min(n0, J - 1);  // rows >= J of a head's score tile are masked below: any finite operand will do
  // a work item = half a clip = 7 wave tiles: with NW = 7 one per wave; with fewer waves a wave walks tiles wave, wave + NW, ...
  auto load_x = [&](int b, int tile, float* x) {
    const int v = tile * 32 + n0;
    const int vc = v < NV ? v : NV - 1;
    const long long tok = (long long)b * NV + vc;
    if (xq) {
      load_slots(xq + tok * 64, x, hb);
    } else {
      const float* p = vt + tok * 3;
      const float p0 = p[0], p1 = p[1], p2 = p[2];
      load_slots(Eq + (long long)vc * 64, x, hb);
#pragma unroll
      for (int s = 0; s < 32; ++s) {
        const int c = slot_channel(s, hb);
        x[s] = (Wv3[c * 3] * p0 + Wv3[c * 3 + 1] * p1 + Wv3[c * 3 + 2] * p2) + x[s];
      }
    }
  };
  auto compute = [&](int b, int tile, float* x) {
    const int v = tile * 32 + n0;
    const bool valid = v < NV;
    const long long tok = (long long)b * NV + (valid ? v : NV - 1);
    // ---- cross-attention (vertex_ca_kernel) ----
    if constexpr (!F16) {
      float n[32];
      {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) s += x[i];
        const float mean = pair_sum(s) * (1.0f / 64.0f);
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
          const float d = x[i] - mean;
          ss += d * d;
        }
        const float inv = 1.0f / (sqrtf(pair_sum(ss) * (1.0f / 63.0f)) + 1e-6f);
#pragma unroll
        for (int i = 0; i < 32; ++i) n[i] = (x[i] - mean) * inv;
      }
      f32x16 sc[2];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[h][r] = sS0[h * 32 + (r & 3) + 8 * (r >> 2) + 4 * hb];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(sK + (nt * J + krow) * LDW64 + 8 * q + 4 * hb);
          sc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, n[4 * q + 0], sc[nt], 0, 0, 0);
          sc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, n[4 * q + 1], sc[nt], 0, 0, 0);
          sc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, n[4 * q + 2], sc[nt], 0, 0, 0);
          sc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, n[4 * q + 3], sc[nt], 0, 0, 0);
        }
      }
      float p[32];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float m = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = (r & 3) + 8 * (r >> 2) + 4 * hb;
          const float sv = (i < J) ? sc[h][r] : -INFINITY;
          sc[h][r] = sv;
          m = fmaxf(m, sv);
        }
        m = pair_max(m);
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float e = __builtin_amdgcn_exp2f(sc[h][r] - m);  // scores are in log2 units (ca_fold)
          p[16 * h + r] = e;
          sum += e;
        }
        const float inv = 1.0f / pair_sum(sum);
#pragma unroll
        for (int r = 0; r < 16; ++r) p[16 * h + r] *= inv;
      }
      f32x16 o[2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[nt][r] = bp[nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hb];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if ((q & 3) < 3 && 8 * (q & 3) < J) {  // wave-uniform: key groups beyond J carry P = 0 (J <= 23: at most 3 per head)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            const f32x4 w =
                *reinterpret_cast<const f32x4*>(sV + (nt * 32 + n0) * CAM_VLD + (q >> 2) * 24 + 8 * (q & 3) + 4 * hb);
            o[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, p[4 * q + 0], o[nt], 0, 0, 0);
            o[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, p[4 * q + 1], o[nt], 0, 0, 0);
            o[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, p[4 * q + 2], o[nt], 0, 0, 0);
            o[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, p[4 * q + 3], o[nt], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) x[16 * nt + r] = x[16 * nt + r] + o[nt][r];  // F1 (stays in registers)
    } else {
      // The same two contractions in the three-product f16 form (round 5): 24 + 12..24 matrix instructions of 32 cycles instead of 64 + 48 of
      // 64.  Both register-side operands carry 2^10 (n: |n| < 8; P: <= 1) so that their lo planes are normal f16 numbers, the LDS-side
      // operands are ca_fold's image (hi | lo planes of Kf 2^sK and Vf 2^sV); one accumulator per product, as in vertex_sa.
      const float kscale = sS0[64], vscale = sS0[65];
      tl_f16x8 nhi[4], nlo[4];
      {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) s += x[i];
        const float mean = pair_sum(s) * (1.0f / 64.0f);
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
          const float d = x[i] - mean;
          ss += d * d;
        }
        const float inv = (1.0f / (sqrtf(pair_sum(ss) * (1.0f / 63.0f)) + 1e-6f)) * 1024.0f;
        float n[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) n[i] = pinned((x[i] - mean) * inv);  // ONE fp32 value for both planes (common.hpp)
#pragma unroll
        for (int sI = 0; sI < 4; ++sI) split_slots8_plain(n + 8 * sI, nhi[sI], nlo[sI]);
      }
      float p[32];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        f32x16 S;
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = 0.f;
        const float* w = sK + (h * J + krow) * LDW64 + hb * 8;
#pragma unroll
        for (int sI = 0; sI < 4; ++sI) {
          const tl_f16x8 whi = *reinterpret_cast<const tl_f16x8*>(w + sI * 16), wlo = *reinterpret_cast<const tl_f16x8*>(w + sI * 16 + 4);
          S = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, nhi[sI], S, 0, 0, 0);
          S = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo, nhi[sI], S, 0, 0, 0);
          S = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, nlo[sI], S, 0, 0, 0);
        }
        float m = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = (r & 3) + 8 * (r >> 2) + 4 * hb;
          const float sv = (i < J) ? fmaf(S[r], kscale, sS0[h * 32 + i]) : -INFINITY;
          S[r] = sv;
          m = fmaxf(m, sv);
        }
        m = pair_max(m);
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float e = __builtin_amdgcn_exp2f(S[r] - m);  // scores are in log2 units (ca_fold)
          p[16 * h + r] = e;
          sum += e;
        }
        const float inv = (1.0f / pair_sum(sum)) * 1024.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) p[16 * h + r] = pinned(p[16 * h + r] * inv);
      }
      f32x16 o[2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[nt][r] = 0.f;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        tl_f16x8 phi, plo;
        split_slots8_plain(p + 16 * h, phi, plo);  // keys 0 .. 15 of the head
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const float* v = sV + (nt * 32 + n0) * CAM_VLD + h * 24 + hb * 8;
          const tl_f16x8 vhi = *reinterpret_cast<const tl_f16x8*>(v), vlo = *reinterpret_cast<const tl_f16x8*>(v + 4);
          o[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vhi, phi, o[nt], 0, 0, 0);
          o[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vlo, phi, o[nt], 0, 0, 0);
          o[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vhi, plo, o[nt], 0, 0, 0);
        }
        if (J > 16) {  // keys 16 .. 23 (wave-uniform): the fragment's upper four elements (keys 24 ..) are zero on both sides
          split_slots8_plain(p + 16 * h + 8, phi, plo);
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            const float* v = sV + (nt * 32 + n0) * CAM_VLD + h * 24 + 16 + hb * 4;
            const tl_f16x4 a4 = *reinterpret_cast<const tl_f16x4*>(v), b4 = *reinterpret_cast<const tl_f16x4*>(v + 2);
            const _Float16 z = (_Float16)0.f;
            const tl_f16x8 vhi = {a4[0], a4[1], a4[2], a4[3], z, z, z, z}, vlo = {b4[0], b4[1], b4[2], b4[3], z, z, z, z};
            o[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vhi, phi, o[nt], 0, 0, 0);
            o[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vlo, phi, o[nt], 0, 0, 0);
            o[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vhi, plo, o[nt], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          x[16 * nt + r] = x[16 * nt + r] + fmaf(o[nt][r], vscale, bp[nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hb]);  // F1 (stays in registers)
    }
    // ---- FFN (adaln_mlp_kernel) ----
    float a[32];
    adaln_slots(x, a, GB + (long long)b * gb_stride + inst * 128, hb);
    f32x16 acc2[2];
    ffn_slots<F16>(sW1, sW2, sB1, sB2, sSc, a, x, acc2, n0, hb);
    if (valid) {
      float y[32];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) y[16 * nt + r] = acc2[nt][r];
      store_slots(yout + tok * 64, y, hb);
    }
  };
  for (int item = blockIdx.x; item < 2 * B; item += gridDim.x) {
    const int b = item >> 1;
    const int tile0 = (item & 1) * 7 + wave;
    // this wave's first 32 query tokens before anything else: their latency hides under the staging of the clip's folded operands
    float x[32];
    load_x(b, tile0, x);
    if (item == (int)blockIdx.x) lds_dma_wait();   // first item: the FFN image's LDS-DMA has landed (wave-uniform)
    __syncthreads();  // the previous item's operands are no longer read (and, first time, the weights are staged)
    if constexpr (F16) {
      // ca_fold's image, a plain copy; every load in flight before the first LDS write
      constexpr int NT = 64 * NW, CIT = (CA_IMG_FLOATS(23) / 4 + NT - 1) / NT;
      const f32x4* src = reinterpret_cast<const f32x4*>(ca_img + (long long)b * CA_IMG_STRIDE);
      const int n4 = CA_IMG_FLOATS(J) / 4;
      f32x4 creg[CIT];
#pragma unroll
      for (int u = 0; u < CIT; ++u) {
        const int i = tid + u * NT;
        if (i < n4) creg[u] = src[i];
      }
#pragma unroll
      for (int u = 0; u < CIT; ++u) {
        const int i = tid + u * NT;
        if (i < n4) reinterpret_cast<f32x4*>(sS0)[i] = creg[u];
      }
    } else {
      const float* Kb = Kf + (long long)b * 4096;
      const float* Vb = Vf + (long long)b * 4096;
      // every load of the item's operands is in flight before the first LDS write (left as plain loops hipcc emits load -> wait ->
      // store per iteration: four exposed L2 round trips per item)
      constexpr int NT = 64 * NW, KIT = (2 * 23 * 16 + NT - 1) / NT, VIT = (64 * 12 + NT - 1) / NT;
      f32x4 kreg[KIT], vreg[VIT];
#pragma unroll
      for (int u = 0; u < KIT; ++u) {  // Kf rows (h, i < J), 16 float4 each
        const int i = tid + u * NT;
        if (i < 2 * J * 16) {
          const int r = i >> 4, c4 = i & 15;
          const int h = r / J, ii = r - h * J;
          kreg[u] = *reinterpret_cast<const f32x4*>(Kb + (h * 32 + ii) * 64 + 4 * c4);
        }
      }
#pragma unroll
      for (int u = 0; u < VIT; ++u) {  // Vf row c: keys 0..23 of both heads, 12 float4
        const int i = tid + u * NT;
        if (i < 64 * 12) {
          const int c = i / 12, k4 = i - c * 12;
          const int h = k4 / 6, j4 = k4 - h * 6;
          vreg[u] = *reinterpret_cast<const f32x4*>(Vb + c * 64 + h * 32 + 4 * j4);
        }
      }
      if (tid < 64) sS0[tid] = s0[(long long)b * 64 + tid];
#pragma unroll
      for (int u = 0; u < KIT; ++u) {
        const int i = tid + u * NT;
        if (i < 2 * J * 16) *reinterpret_cast<f32x4*>(sK + (i >> 4) * LDW64 + 4 * (i & 15)) = kreg[u];
      }
#pragma unroll
      for (int u = 0; u < VIT; ++u) {
        const int i = tid + u * NT;
        if (i < 64 * 12) {
          const int c = i / 12, k4 = i - c * 12;
          const int h = k4 / 6, j4 = k4 - h * 6;
          *reinterpret_cast<f32x4*>(sV + c * CAM_VLD + h * 24 + 4 * j4) = vreg[u];
        }
      }
    }
    __syncthreads();
    compute(b, tile0, x);
    if (NW < 7) {
      for (int tw = wave + NW; tw < 7; tw += NW) {
        load_x(b, (item & 1) * 7 + tw, x);
        compute(b, (item & 1) * 7 + tw, x);
      }
    }
  }
}

// ======================================================================================================
// adaln_qkv: qkv[tok][0:192] = Wqkv * AdaLN(x) + bqkv   (vertex self-attention input, CoevoDecoder.py:103,120)
// ======================================================================================================
// (fp32 pipe; the split mode's qkv product lives inside vertex_sab, from the weight image of pmce_qkv_pack_f16: stage_weight_split's
// layout of Wqkv * 2^s as (hi | lo) f16 fragments, then {2^s, 2^-s})
#define QKV_IMG_FLOATS (192 * LDW64 + 32)
__global__ __launch_bounds__(256, 1) void adaln_qkv_kernel(const float* __restrict__ xin, const float* __restrict__ GB,
                                                        int gb_stride, int inst, const float* __restrict__ Wqkv,
                                                        const float* __restrict__ bqkv, float* __restrict__ qkv, int B) {
  __shared__ __attribute__((aligned(16))) float sW[192 * LDW64];
  __shared__ __attribute__((aligned(16))) float sT[4 * 32 * 32];  // per-wave output transpose tile: [token][8 chunks of 4 channels], chunk ^ (token & 7)
  __shared__ float sB[192];
  const int tid = threadIdx.x;
  stage_weight<64>(sW, Wqkv, 192, tid, 256);
  if (tid < 192) sB[tid] = bqkv[tid];
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  const int n0 = lane & 31, hb = lane >> 5;
  const int ntiles = B * NTILE;
  // The kernel is latency-bound (PMC: 44 % of its wave cycles parked in s_waitcnt, matrix pipe 30 % busy): a wave's next
  // tile is fetched while the current one is in the matrix pipe.
  auto tok_of = [&](int wt) {
    const int v = (wt % NTILE) * 32 + n0;
    return (long long)(wt / NTILE) * NV + (v < NV ? v : NV - 1);
  };
  float x[32];
  int wt = blockIdx.x * 4 + wave;
  if (wt < ntiles) load_slots(xin + tok_of(wt) * 64, x, hb);
  for (; wt < ntiles; wt += gridDim.x * 4) {
    const int b = wt / NTILE, tile = wt % NTILE;
    float a[32];
    adaln_slots(x, a, GB + (long long)b * gb_stride + inst * 128, hb);
    const int wn = wt + gridDim.x * 4;
    if (wn < ntiles) load_slots(xin + tok_of(wn) * 64, x, hb);  // x is dead after the AdaLN: its registers take the next tile
    // Output through a wave-private LDS tile so that every store instruction writes full 128-byte lines (8 lanes per token
    // and 32-channel group) instead of 32 bytes in each of 32 rows: the accumulator layout's direct stores cost 24 of this
    // kernel's 59 us.  The 192 output channels are produced in two halves of three 32-channel tiles (48 instead of 96
    // accumulator registers: room for the prefetched tile without spilling).
    // (rows of 128 B with the 16-byte chunk XOR-ed by the row: the 8 consecutive lanes of a ds_write_b128 service group - 8 tokens,
    // one chunk - hit 8 different bank quads, and the four non-contiguous 16-lane groups of the ds_read_b128 - 4 rows, half a row each -
    // cover the 256-byte bank row exactly once; the padded 36-float rows this replaces cost 331,776 conflict cycles per launch)
    float* tb = sT + wave * (32 * 32);
    const long long tok0 = (long long)b * NV + tile * 32;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      f32x16 acc[3];
#pragma unroll
      for (int nt = 0; nt < 3; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = sB[(3 * half + nt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hb];
      tl_gemm<8, 3, LDW64>(sW + 3 * half * 32 * LDW64, a, acc, n0, hb);
#pragma unroll
      for (int nt = 0; nt < 3; ++nt) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 t;
          t.x = acc[nt][4 * g + 0];
          t.y = acc[nt][4 * g + 1];
          t.z = acc[nt][4 * g + 2];
          t.w = acc[nt][4 * g + 3];
          *reinterpret_cast<f32x4*>(tb + n0 * 32 + 4 * ((2 * g + hb) ^ (n0 & 7))) = t;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int tk = 8 * i + (lane >> 3);
          const f32x4 t = *reinterpret_cast<const f32x4*>(tb + tk * 32 + 4 * ((lane & 7) ^ (tk & 7)));
          if (tile * 32 + tk < NV)
            *reinterpret_cast<f32x4*>(qkv + (tok0 + tk) * 192 + (3 * half + nt) * 32 + 4 * (lane & 7)) = t;
        }
      }
    }
  }
}

// ======================================================================================================
// vertex_sa (fp32 pipe): y = x + proj(softmax(q k^T / sqrt(32)) v), 2 heads x 32, 431 x 431 per clip (CoevoDecoder.py:118-131).
// grid (2, B) x 448 threads: wave w owns query tile blockIdx.x*7+w for BOTH heads; the 14 key tiles (32 keys x
// {k,v} x 64 ch) stream through a double-buffered LDS ring shared by the 7 waves.  S^T = K Q^T puts one query
// per lane pair, so the online softmax is in-lane; P^T is reused directly as the B operand of O^T += V^T P^T.
// (The split mode runs vertex_sab below: the same attention in the three-product f16 form, fused with its qkv product.)
// ======================================================================================================
#define SA_KLD 68
#define SA_VLD 64
#define SA_VTLD 36  // vertex_sab: V^T rows (one channel, 32 keys as 2 k-steps x 2 lane halves x (hi | lo) x 8 f16 = 32 floats) + 4
__global__ __launch_bounds__(448) void vertex_sa_kernel(const float* __restrict__ xin, const float* __restrict__ qkv,
                                                        const float* __restrict__ Wp, const float* __restrict__ bp,
                                                        float* __restrict__ yout) {
  __shared__ __attribute__((aligned(16))) float sK[2][32 * SA_KLD];
  __shared__ __attribute__((aligned(16))) float sVv[2][32 * SA_VLD];
  __shared__ __attribute__((aligned(16))) float sWp[64 * LDW64];
  const int b = blockIdx.y, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int n0 = lane & 31, hb = lane >> 5;
  const float* qkv_b = qkv + (long long)b * NV * 192;
  stage_weight<64>(sWp, Wp, 64, tid, 448);

  // staging: 32 rows x 32 float4 (16 of k, 16 of v) = 1024 float4 per key tile, thread -> (row, float4) in order
  f32x4 pre[3];
  auto gload = [&](int jt) {
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      const int idx = tid + it * 448;
      if (idx < 1024) {
        const int rr = idx >> 5, c4 = idx & 31;
        const int j = jt * 32 + rr;
        pre[it] = (j < NV) ? *reinterpret_cast<const f32x4*>(qkv_b + (long long)j * 192 + 64 + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      const int idx = tid + it * 448;
      if (idx < 1024) {
        const int rr = idx >> 5, c4 = idx & 31;
        if (c4 < 16)
          *reinterpret_cast<f32x4*>(&sK[buf][rr * SA_KLD + 4 * c4]) = pre[it];
        else
          *reinterpret_cast<f32x4*>(&sVv[buf][rr * SA_VLD + 4 * (c4 - 16)]) = pre[it];
      }
    }
  };

  const int qtile = blockIdx.x * 7 + wave;
  const int v = qtile * 32 + n0;
  const bool valid = v < NV;
  const long long tok = (long long)b * NV + (valid ? v : NV - 1);
  float q[32];
  load_slots(qkv + tok * 192, q, hb);
  // 32^-0.5 * log2(e): scores are kept in log2 units so the softmax uses the native v_exp_f32 (2^x)
  const float scale = 0.17677669529663688110f * 1.44269504088896340736f;
#pragma unroll
  for (int s = 0; s < 32; ++s) q[s] = pinned(q[s] * scale);

  f32x16 O[2];
  float mrun[2], lrun[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    mrun[h] = -INFINITY;
    lrun[h] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) O[h][r] = 0.f;
  }
  // tile jt + 1 is written to LDS at the TOP of iteration jt (from the registers iteration jt - 1 loaded), tile jt + 2 is fetched
  // right after: the staging runs under the fragment reads' latency instead of in front of the barrier
  gload(0);
  lstore(0);
  if (NTILE > 1) gload(1);
  __syncthreads();
  for (int jt = 0; jt < NTILE; ++jt) {
    const int buf = jt & 1;
    if (jt + 1 < NTILE) lstore(buf ^ 1);
    if (jt + 2 < NTILE) gload(jt + 2);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x16 S;
#pragma unroll
      for (int r = 0; r < 16; ++r) S[r] = 0.f;
      tl_gemm<4, 1, SA_KLD>(&sK[buf][32 * h], q + 16 * h, &S, n0, hb);
      float mt = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = jt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hb;
        const float sv = (j < NV) ? S[r] : -INFINITY;
        S[r] = sv;
        mt = fmaxf(mt, sv);
      }
      mt = pair_max(mt);
      const float mn = fmaxf(mrun[h], mt);
      const float corr = __builtin_amdgcn_exp2f(mrun[h] - mn);
      float pr[16], sum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        pr[r] = __builtin_amdgcn_exp2f(S[r] - mn);
        sum += pr[r];
      }
      lrun[h] = lrun[h] * corr + pair_sum(sum);
      mrun[h] = mn;
#pragma unroll
      for (int r = 0; r < 16; ++r) O[h][r] *= corr;
      const float* vb = &sVv[buf][4 * hb * SA_VLD + 32 * h + n0];
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const float a = vb[((s & 3) + 8 * (s >> 2)) * SA_VLD];
        O[h] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, pr[s], O[h], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  float att[32];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float inv = 1.0f / lrun[h];
#pragma unroll
    for (int r = 0; r < 16; ++r) att[16 * h + r] = O[h][r] * inv;
  }
  float x[32];
  load_slots(xin + tok * 64, x, hb);
  f32x16 acc[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nt][r] = bp[nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hb] + x[16 * nt + r];
  tl_gemm<8, 2, LDW64>(sWp, att, acc, n0, hb);
  if (valid) {
    float y[32];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) y[16 * nt + r] = acc[nt][r];
    store_slots(yout + tok * 64, y, hb);
  }
}

// ======================================================================================================
// vertex_sab - the attention half of the vertex stream's AdaLN Block in ONE launch (split mode):
//   y = x + proj(softmax(q k^T / sqrt(32)) v),   [q | k | v] = Linear(64 -> 192)(AdaLN(x))      (CoevoDecoder.py:103, :118-131)
// 431 x 431 per clip, 2 heads x 32, flash-style; every contraction but the 64 x 64 output projection in the three-product f16 form.
// Phase 1: every wave runs AdaLN + the 64 -> 192 product (the weight's image of pmce_qkv_pack_f16, staged by LDS-DMA) for two 32-token
//   tiles.  The accumulator layouts ARE the operand layouts of phase 2, so nothing is transposed or re-laid-out:
//     q (swapped product, lane = token, slot order of the head's channels)  = the B fragments of S^T = K Q^T: split in registers;
//     k (swapped product)                                                   = the A fragments of S^T (lane = key);
//     v (operands exchanged: D[token][channel], lane = channel, r = key)     = the A fragments of O^T += V^T P^T (lane = channel, 8 keys).
//   k and v leave as f16 (hi | lo) fragment planes for a per-clip scratch (14 key tiles x 16 KB, written and read by this workgroup only:
//   it never leaves the L2) - the whole clip's keys do not fit in LDS beside the weight image (224 KB).  Every value is pinned to ONE
//   fp32 number before it is split (common.hpp).  (Until round 5 two launches - adaln_qkv with a [B,431,192] fp32 QKV round trip, then
//   the attention, whose staging waves split and transposed K / V on their way into LDS: the fused kernel reproduced that form bit for
//   bit, profiles/r05_a_*, before it was removed.)
// Phase 2: the 14 key tiles stream through a double-buffered LDS ring shared by the 7 waves (a plain 16 KB copy per tile).  S^T = K Q^T
//   puts one query per lane pair, so the online softmax is in-lane, and P^T in the accumulator layout is directly the B operand of
//   O^T += V^T P^T.  One accumulator per product: the lo planes are kept at their true magnitude and the register-side operands carry a
//   power of two that keeps THEIR lo halves normal f16 numbers (q: 2^10, undone inside the exp2's multiply-add; P: 2^6, cancelled by 1/l).
//   The lo halves of small K / V elements (|x| < 0.25) are subnormal f16; the matrix pipe reads subnormals as they are
//   (scripts/microbench/mfma_denorm.hip, test_mfma_reads_f16_subnormals), so each costs at most 2^-25 absolute.  24 matrix instructions
//   of 32 cycles per key tile and query tile.  The softmax rescales lazily: the reference maximum only moves (a wave-uniform branch)
//   when some query's tile maximum exceeds it by more than 2^8 - same mathematics, P <= 2^14 in f16.
// QT = 2: one workgroup per clip, two query tiles per wave (B > 128: staging, barrier and fragment reads are paid per (workgroup, key
//   tile) whatever the number of queries a wave owns, and 2 B workgroups would be two rounds).  QT = 1: two workgroups per clip, one
//   query tile per wave; both compute all 14 key tiles (into their own scratch halves): at B <= 128 the second workgroup runs on a CU
//   that would idle.  Same arithmetic per query tile in the same order: a clip's result does not depend on the batch it came in.
// ======================================================================================================
#define SAB_TILE_FLOATS 4096  // one key tile in scratch: K [32 keys][64 floats] then V^T [64 channels][32 floats]
template <int QT>
__global__ __launch_bounds__(448) void vertex_sab_kernel(const float* __restrict__ xin, const float* __restrict__ GB, int gb_stride, int inst,
                                                         const float* __restrict__ qkv_img, const float* __restrict__ bqkv,
                                                         const float* __restrict__ Wp, const float* __restrict__ bp, float* kvs,
                                                         float* __restrict__ yout) {
  constexpr int G = 2 / QT;  // workgroups per clip
  __shared__ __attribute__((aligned(16))) float sW[192 * LDW64];  // phase 1: the qkv weight's f16 image; afterwards rows 0 .. 63: Wproj (fp32)
  __shared__ __attribute__((aligned(16))) float sK[2][32 * SA_KLD];
  __shared__ __attribute__((aligned(16))) float sVv[2][64 * SA_VTLD];
  __shared__ __attribute__((aligned(16))) float sQ[QT == 2 ? 7 * 8 * 64 * 4 : 4];  // the second query tile's 8 q fragments per wave
  __shared__ float sB[192];
  __shared__ __attribute__((aligned(16))) float sGB[128];  // the clip's AdaLN gamma | beta (every tile of the workgroup uses them)
  __shared__ float sSc[2];
  const int b = blockIdx.y, g = blockIdx.x, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int n0 = lane & 31, hb = lane >> 5;
  static_assert(192 * LDW64 * 4 % 1024 == 0, "the qkv weight image is a whole number of KiB");
  lds_dma_copy(sW, qkv_img, 192 * LDW64 * 4 / 1024, wave, 7, lane);  // 51 KiB, every piece in flight at once
  if (tid < 2) sSc[tid] = qkv_img[192 * LDW64 + tid];
  if (tid < 192) sB[tid] = bqkv[tid];
  if (tid >= 192 && tid < 320) sGB[tid - 192] = GB[(long long)b * gb_stride + inst * 128 + tid - 192];
  float* kvw = kvs + (size_t)(b * G + g) * (NTILE * SAB_TILE_FLOATS);  // this workgroup's scratch (plain pointer: written, then read)
  const float* gb = sGB;

  // ---- phase 1 -----------------------------------------------------------------------------------------------------------
  // one 32 x 32 tile of the 64 -> 192 product: nt = output channels 32 nt .. +31; exchanged = operands swapped (D[token][channel])
  auto product = [&](const tl_f16x8(&ahi)[4], const tl_f16x8(&alo)[4], int nt, bool exchanged, f32x16& m, f32x16& c) {
#pragma unroll
    for (int r = 0; r < 16; ++r) m[r] = c[r] = 0.f;
    const float* w = sW + (nt * 32 + n0) * LDW64 + hb * 8;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const tl_f16x8 whi = *reinterpret_cast<const tl_f16x8*>(w + s * 16), wlo = *reinterpret_cast<const tl_f16x8*>(w + s * 16 + 4);
      if (!exchanged) {
        m = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, ahi[s], m, 0, 0, 0);
        m = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo, ahi[s], m, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, alo[s], c, 0, 0, 0);
      } else {
        m = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[s], whi, m, 0, 0, 0);
        m = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[s], wlo, m, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[s], whi, c, 0, 0, 0);
      }
    }
  };
  // AdaLN of key/query tile T (its 32 tokens, clamped at the clip's end) -> the product's activation fragments
  auto activation = [&](const float* x, tl_f16x8(&ahi)[4], tl_f16x8(&alo)[4]) {
    float a[32];
    adaln_slots(x, a, gb, hb);
#pragma unroll
    for (int s = 0; s < 4; ++s) split_slots8(a + 8 * s, ahi[s], alo[s]);
  };
  auto tok_of = [&](int T) {
    const int v = T * 32 + n0;
    return (long long)b * NV + (v < NV ? v : NV - 1);
  };
  // 32^-0.5 * log2(e): scores in log2 units so that the softmax uses the native v_exp_f32 (2^x); times the 2^10 of q's f16 split
  const float scale = 0.17677669529663688110f * 1.44269504088896340736f * 1024.0f;
  tl_f16x8 qhi[2][2], qlo[2][2];  // query tile 0: [head][k-step]
  tl_f16x8* sQw = reinterpret_cast<tl_f16x8*>(sQ) + wave * 8 * 64 + lane;  // (QT = 2) fragment f of this lane: sQw[f * 64]
  // q of one tile from its activation fragments: t = 0 -> registers, t = 1 -> LDS
  auto make_q = [&](const tl_f16x8(&ahi)[4], const tl_f16x8(&alo)[4], int t, float down) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x16 m, c;
      product(ahi, alo, h, false, m, c);
      float q[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = pinned(fmaf(fmaf(c[r], 0.00048828125f, m[r]), down, sB[h * 32 + (r & 3) + 8 * (r >> 2) + 4 * hb]));
        q[r] = pinned(v * scale);  // ONE fp32 value for both planes (common.hpp)
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (t == 0) {
          split_slots8_plain(q + 8 * ks, qhi[h][ks], qlo[h][ks]);
        } else {
          tl_f16x8 fh, fl;
          split_slots8_plain(q + 8 * ks, fh, fl);
          sQw[((h * 2 + ks) * 2 + 0) * 64] = fh;
          sQw[((h * 2 + ks) * 2 + 1) * 64] = fl;
        }
      }
    }
  };
  {
    float x0[32], x1[32];
    load_slots(xin + tok_of(2 * wave) * 64, x0, hb);  // both tiles requested before the weight image's barrier
    load_slots(xin + tok_of(2 * wave + 1) * 64, x1, hb);
    lds_dma_wait();
    __syncthreads();  // the weight image, sB, sSc
    const float down = sSc[1];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int T = 2 * wave + t;
      const bool tok_valid = T * 32 + n0 < NV;
      tl_f16x8 ahi[4], alo[4];
      activation(t == 0 ? x0 : x1, ahi, alo);
      float* kt = kvw + (size_t)T * SAB_TILE_FLOATS;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        f32x16 m, c;
        product(ahi, alo, 2 + h, false, m, c);  // k: lane = key, registers = the head's channels in slot order
        float kv_[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = pinned(fmaf(fmaf(c[r], 0.00048828125f, m[r]), down, sB[64 + h * 32 + (r & 3) + 8 * (r >> 2) + 4 * hb]));
          kv_[r] = tok_valid ? v : 0.f;  // keys beyond the clip: zero rows (their scores are masked)
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          tl_f16x8 fh, fl;
          split_slots8_plain(kv_ + 8 * ks, fh, fl);
          float* d = kt + n0 * 64 + ((h * 2 + ks) * 2 + hb) * 8;
          *reinterpret_cast<tl_f16x8*>(d) = fh;
          *reinterpret_cast<tl_f16x8*>(d + 4) = fl;
        }
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        f32x16 m, c;
        product(ahi, alo, 4 + h, true, m, c);  // v, operands exchanged: lane = channel 32 h + n0, register r = key (r & 3) + 8 (r >> 2) + 4 hb
        const float bias = sB[128 + h * 32 + n0];
        float vt_[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = pinned(fmaf(fmaf(c[r], 0.00048828125f, m[r]), down, bias));
          vt_[r] = (T * 32 + (r & 3) + 8 * (r >> 2) + 4 * hb < NV) ? v : 0.f;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          tl_f16x8 fh, fl;
          split_slots8_plain(vt_ + 8 * ks, fh, fl);
          float* d = kt + 2048 + (32 * h + n0) * 32 + (ks * 2 + hb) * 8;
          *reinterpret_cast<tl_f16x8*>(d) = fh;
          *reinterpret_cast<tl_f16x8*>(d + 4) = fl;
        }
      }
      if (QT == 2) make_q(ahi, alo, t, down);
    }
    if (QT == 1) {  // this wave's query tile is in general not one of its two key tiles
      float xq_[32];
      load_slots(xin + tok_of(g * 7 + wave) * 64, xq_, hb);
      tl_f16x8 ahi[4], alo[4];
      activation(xq_, ahi, alo);
      make_q(ahi, alo, 0, down);
    }
  }
  __syncthreads();  // every key tile of the clip is in the scratch (and visible to the workgroup); the weight image is dead

  // ---- phase 2: the key loop ---------------------------------------------------------------------------------------------------
  float* sWp = sW;
  stage_weight<64>(sWp, Wp, 64, tid, 448);
  f32x4 pre[3];  // a key tile = 1024 float4 over 448 threads
  auto gload = [&](int jt) {
    const float* src = kvw + (size_t)jt * SAB_TILE_FLOATS;
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      const int idx = tid + it * 448;
      if (idx < 1024) pre[it] = *reinterpret_cast<const f32x4*>(src + 4 * idx);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      const int idx = tid + it * 448;
      if (idx < 512) {
        *reinterpret_cast<f32x4*>(&sK[buf][(idx >> 4) * SA_KLD + 4 * (idx & 15)]) = pre[it];
      } else if (idx < 1024) {
        const int i2 = idx - 512;
        *reinterpret_cast<f32x4*>(&sVv[buf][(i2 >> 3) * SA_VTLD + 4 * (i2 & 7)]) = pre[it];
      }
    }
  };
  bool valid[QT];
  long long tok[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const int v = ((QT == 2 ? 2 * wave : g * 7 + wave) + t) * 32 + n0;
    valid[t] = v < NV;
    tok[t] = (long long)b * NV + (valid[t] ? v : NV - 1);
  }
  f32x16 O[QT][2];
  float mrun[QT][2], lrun[QT][2], off[QT][2];
#pragma unroll
  for (int t = 0; t < QT; ++t)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      mrun[t][h] = -INFINITY;
      lrun[t][h] = 0.f;
      off[t][h] = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) O[t][h][r] = 0.f;
    }
  constexpr float kQs = 0.0009765625f;
  constexpr float kLazy = 8.0f * 1024.0f;
  gload(0);
  lstore(0);
  if (NTILE > 1) gload(1);
  __syncthreads();
  for (int jt = 0; jt < NTILE; ++jt) {
    const int buf = jt & 1;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      tl_f16x8 kf[2][2], vf[2][2];  // [k-step][hi | lo]
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const float* kp = &sK[buf][n0 * SA_KLD + ((h * 2 + ks) * 2 + hb) * 8];
        kf[ks][0] = *reinterpret_cast<const tl_f16x8*>(kp);
        kf[ks][1] = *reinterpret_cast<const tl_f16x8*>(kp + 4);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const float* vp = &sVv[buf][(32 * h + n0) * SA_VTLD + (ks * 2 + hb) * 8];
        vf[ks][0] = *reinterpret_cast<const tl_f16x8*>(vp);
        vf[ks][1] = *reinterpret_cast<const tl_f16x8*>(vp + 4);
      }
      if (h == 0) {
        asm volatile("" ::: "memory");
        if (jt + 1 < NTILE) lstore(buf ^ 1);
        if (jt + 2 < NTILE) gload(jt + 2);
      }
#pragma unroll
      for (int t = 0; t < QT; ++t) {
        f32x16 S;
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const tl_f16x8 qh = t == 0 ? qhi[h][ks] : sQw[((h * 2 + ks) * 2 + 0) * 64];
          const tl_f16x8 ql = t == 0 ? qlo[h][ks] : sQw[((h * 2 + ks) * 2 + 1) * 64];
          S = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks][0], qh, S, 0, 0, 0);
          S = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks][1], qh, S, 0, 0, 0);
          S = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks][0], ql, S, 0, 0, 0);
        }
        float mt = -INFINITY;
        if (jt == NTILE - 1) {  // only the clip's last key tile holds keys beyond it (wave-uniform)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int j = jt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hb;
            S[r] = (j < NV) ? S[r] : -INFINITY;
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) mt = fmaxf(mt, S[r]);
        if (__builtin_amdgcn_ballot_w64(mt > mrun[t][h] + kLazy) != 0) {
          const float mn = fmaxf(mrun[t][h], pair_max(mt));
          const float corr = __builtin_amdgcn_exp2f((mrun[t][h] - mn) * kQs);
          mrun[t][h] = mn;
          off[t][h] = fmaf(mn, -kQs, 6.0f);
          lrun[t][h] *= corr;
#pragma unroll
          for (int r = 0; r < 16; ++r) O[t][h][r] *= corr;
        }
        float pr[16], sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          pr[r] = __builtin_amdgcn_exp2f(fmaf(S[r], kQs, off[t][h]));
          sum += pr[r];
        }
        lrun[t][h] += sum;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          tl_f16x8 phi, plo;
          split_slots8_plain(pr + 8 * ks, phi, plo);
          O[t][h] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[ks][0], phi, O[t][h], 0, 0, 0);
          O[t][h] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[ks][1], phi, O[t][h], 0, 0, 0);
          O[t][h] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[ks][0], plo, O[t][h], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    float att[32];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float inv = 1.0f / pair_sum(lrun[t][h]);
#pragma unroll
      for (int r = 0; r < 16; ++r) att[16 * h + r] = O[t][h][r] * inv;
    }
    float x[32];
    load_slots(xin + tok[t] * 64, x, hb);
    f32x16 acc[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] = bp[nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hb] + x[16 * nt + r];
    tl_gemm<8, 2, LDW64>(sWp, att, acc, n0, hb);
    if (valid[t]) {
      float y[32];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) y[16 * nt + r] = acc[nt][r];
      store_slots(yout + tok[t] * 64, y, hb);
    }
  }
}

// ======================================================================================================
// tokens_kv (joint<-vertex direction, live in coevoblock3 only):
//   kv[tok][0:64]   = Wk * AdaLN_k(xk) + bk     xk = proj_v2j_dim(vf) + v2j_K_embed   (CoevoDecoder.py:183)
//   kv[tok][64:128] = Wv * AdaLN_v(xv) + bv     xv = vf
// Generic form: xk/xv given ([B,431,64]).  Fused form (xk == nullptr): vf = Wv3*vt + Ev[v] formed on the fly
// (Ev = vertx_proj.bias + vertx_pos_embed) and xk = Wv2j*vf + Ek[v] (Ek = proj_v2j_dim.bias + v2j_K_embed).
// F16 (split mode, round 5): the three 64 x 64 products in the three-product f16 form of the FFN kernels - 72 matrix instructions of 32
// cycles per 32-token tile instead of 192 of 64 - from an image of the three weights made once (pmce_tkv_pack_f16: per weight
// stage_weight_split's rows of W * 2^s, then the three {2^s, 2^-s} pairs), staged by LDS-DMA; with that the kernel fits two workgroups
// per CU (the fp32 form's 192 64-cycle instructions per tile are scheduled over 380 registers: one wave per SIMD).
// ======================================================================================================
#define TKV_IMG_FLOATS (3 * 64 * LDW64 + 32)
template <bool F16>
__global__ __launch_bounds__(256, F16 ? 2 : 1) void tokens_kv_kernel(const float* __restrict__ xk_in, const float* __restrict__ xv_in,
                                                        const float* __restrict__ vt, const float* __restrict__ Wv3,
                                                        const float* __restrict__ Ev, const float* __restrict__ Wv2j,
                                                        const float* __restrict__ Ek, const float* __restrict__ GB,
                                                        int gb_stride, int ik, int iv, const float* __restrict__ Wk,
                                                        const float* __restrict__ bk, const float* __restrict__ Wv,
                                                        const float* __restrict__ bv, float* __restrict__ kv, int B,
                                                        const float* __restrict__ img) {
  __shared__ __attribute__((aligned(16))) float sW[3 * 64 * LDW64];  // [W2 | Wk | Wv]: fp32 rows, or (F16) the image's f16 planes
  __shared__ float sSc[8];
  float* sW2 = sW;
  float* sWk = sW + 64 * LDW64;
  float* sWv = sW + 2 * 64 * LDW64;
  const int tid = threadIdx.x;
  if constexpr (F16) {
    static_assert(3 * 64 * LDW64 * 4 % 1024 == 0, "the image's planes are a whole number of KiB");
    lds_dma_copy(sW, img, 3 * 64 * LDW64 * 4 / 1024, tid >> 6, 4, tid & 63);
    if (tid < 6) sSc[tid] = img[3 * 64 * LDW64 + tid];
    lds_dma_wait();
  } else {
    stage_weight<64>(sWk, Wk, 64, tid, 256);
    stage_weight<64>(sWv, Wv, 64, tid, 256);
    if (!xk_in) stage_weight<64>(sW2, Wv2j, 64, tid, 256);
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  const int n0 = lane & 31, hb = lane >> 5;
  const int ntiles = B * NTILE;
  // y[s] = bias-or-embedding[s] + (W x)[s] for the 64 x 64 weight at `sw`: fp32 pipe, or the three-product f16 form (scale pair `isc`)
  auto linear64 = [&](const float* sw, int isc, const float* x, const float* add, float* y) {
    if constexpr (F16) {
      const float down = sSc[2 * isc + 1];
      tl_f16x8 xhi[4], xlo[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) split_slots8(x + 8 * s, xhi[s], xlo[s]);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        f32x16 m, c;
#pragma unroll
        for (int r = 0; r < 16; ++r) m[r] = c[r] = 0.f;
        const float* w = sw + (nt * 32 + n0) * LDW64 + hb * 8;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const tl_f16x8 whi = *reinterpret_cast<const tl_f16x8*>(w + s * 16), wlo = *reinterpret_cast<const tl_f16x8*>(w + s * 16 + 4);
          m = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, xhi[s], m, 0, 0, 0);
          m = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo, xhi[s], m, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, xlo[s], c, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) y[16 * nt + r] = fmaf(fmaf(c[r], 0.00048828125f, m[r]), down, add[16 * nt + r]);
      }
    } else {
      f32x16 t[2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) t[nt][r] = add[16 * nt + r];
      tl_gemm<8, 2, LDW64>(sw, x, t, n0, hb);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) y[16 * nt + r] = t[nt][r];
    }
  };
  for (int wt = blockIdx.x * 4 + wave; wt < ntiles; wt += gridDim.x * 4) {
    const int b = wt / NTILE, tile = wt % NTILE;
    const int v = tile * 32 + n0;
    const bool valid = v < NV;
    const int vc = valid ? v : NV - 1;
    const long long tok = (long long)b * NV + vc;
    float xk[32], xv[32];
    if (xk_in) {
      load_slots(xk_in + tok * 64, xk, hb);
      load_slots(xv_in + tok * 64, xv, hb);
    } else {
      const float* p = vt + tok * 3;
      const float p0 = p[0], p1 = p[1], p2 = p[2];
      load_slots(Ev + (long long)vc * 64, xv, hb);
#pragma unroll
      for (int s = 0; s < 32; ++s) {
        const int c = slot_channel(s, hb);
        xv[s] = (Wv3[c * 3] * p0 + Wv3[c * 3 + 1] * p1 + Wv3[c * 3 + 2] * p2) + xv[s];
      }
      float e[32];
      load_slots(Ek + (long long)vc * 64, e, hb);
      linear64(sW2, 0, xv, e, xk);
    }
    const float* gb = GB + (long long)b * gb_stride;
    float a[32], bias[32], y[32];
    // k
    adaln_slots(xk, a, gb + ik * 128, hb);
#pragma unroll
    for (int s = 0; s < 32; ++s) bias[s] = bk[slot_channel(s, hb)];
    linear64(sWk, 1, a, bias, y);
    if (valid) store_slots(kv + tok * 128, y, hb);
    // v
    adaln_slots(xv, a, gb + iv * 128, hb);
#pragma unroll
    for (int s = 0; s < 32; ++s) bias[s] = bv[slot_channel(s, hb)];
    linear64(sWv, 2, a, bias, y);
    if (valid) store_slots(kv + tok * 128 + 64, y, hb);
  }
}
// the three weights' f16 image: [W2j | Wk | Wv] rows as stage_weight_split writes them, then {2^s, 2^-s} per weight
__global__ __launch_bounds__(256) void tkv_pack_kernel(const float* __restrict__ W2, const float* __restrict__ Wk, const float* __restrict__ Wv,
                                                       float* __restrict__ img) {
  __shared__ float red[16];
  const float* W[3] = {W2, Wk, Wv};
  for (int i = 0; i < 3; ++i) {
    float up, down;
    weight_scale(W[i], 64 * 64, red, threadIdx.x, 256, up, down);
    stage_weight_split<64>(img + i * 64 * LDW64, W[i], 64, up, threadIdx.x, 256);
    if (threadIdx.x == 0) {
      img[3 * 64 * LDW64 + 2 * i] = up;
      img[3 * 64 * LDW64 + 2 * i + 1] = down;
    }
  }
}

// ======================================================================================================
// joint_stream: the joint side of coevoblock3, one workgroup per clip (J <= 32 tokens; ~4 MFLOP, VALU).
//   y = xq + proj(softmax(q K^T / sqrt(8)) V) over the 431 vertex keys, 8 heads of 8     (joint_CA_FFN, :83)
//   stage 1 stops here (standalone cross-attention op); stage 4 skips the cross-attention block (self-attention block
//   alone: the reference's Block module).  Otherwise:
//   y += Mlp(AdaLN(y)) ; y += SA(AdaLN(y)) (8 heads over J) ; y += Mlp(AdaLN(y)) ; cam_pose = Wc*y + bc + jt
// ======================================================================================================
struct JointStreamW {
  const float *wq, *bq, *proj_w, *proj_b;                       // joint_CA_FFN.attn
  const float *fc1_w, *fc1_b, *fc2_w, *fc2_b;                   // joint_CA_FFN.mlp
  const float *qkv_w, *qkv_b, *sproj_w, *sproj_b;               // joint_SA_FFN.attn
  const float *sfc1_w, *sfc1_b, *sfc2_w, *sfc2_b;               // joint_SA_FFN.mlp
  const float *coor_w, *coor_b;                                 // proj_joint_feat2coor
  int i_normq, i_norm2, i_snorm1, i_snorm2;                     // AdaLN instance indices into GB
};

__global__ __launch_bounds__(JS_THREADS) void joint_stream_kernel(const float* __restrict__ xq_in, const float* __restrict__ jQ,
                                                                  const float* __restrict__ kv, const float* __restrict__ GB,
                                                                  int gb_stride, JointStreamW w, const float* __restrict__ jt,
                                                                  float* __restrict__ y_out, float* __restrict__ pose_out, int J,
                                                                  int stage) {
  __shared__ float s_y[32 * JLD];
  __shared__ float s_a[32 * JLD];
  __shared__ float s_q[32 * JLD];
  __shared__ __attribute__((aligned(16))) float s_h[32 * 257];   // MLP hidden / qkv scratch (>= 32*193); partial attention states
  __shared__ __attribute__((aligned(16))) float s_kv[64 * 128];  // one tile of 64 vertex keys: k | v
  __shared__ __attribute__((aligned(16))) float s_w[64 * JWL];   // one 64 x 64 weight tile
  __shared__ float s_gb[4 * 128];                                // this clip's gamma | beta of the four AdaLN instances
  __shared__ float s_cw[3 * 64 + 3];                             // proj_joint_feat2coor
  const int b = blockIdx.x, tid = threadIdx.x;
  {  // small per-clip / per-launch operands, fetched once up front instead of at their (latency-exposed) points of use
    const float* gbg = GB + (long long)b * gb_stride;
    const int inst = tid >> 7, c = tid & 127;
    const int ii = inst == 0 ? w.i_normq : inst == 1 ? w.i_norm2 : inst == 2 ? w.i_snorm1 : w.i_snorm2;
    s_gb[inst * 128 + c] = gbg[ii * 128 + c];
    if (tid < 192) s_cw[tid] = w.coor_w[tid];
    if (tid < 3) s_cw[192 + tid] = w.coor_b[tid];
  }
  JTile pre;
  // the first weight tile this launch needs, requested before anything else
  if (stage != 4) jtile_fetch(pre, w.wq, 64, 0, 0, tid);
  else jtile_fetch(pre, w.qkv_w, 64, 0, 0, tid);
  for (int idx = tid; idx < J * 64; idx += JS_THREADS) {
    const int i = idx >> 6, c = idx & 63;
    float v = xq_in[((long long)b * J + i) * 64 + c];
    if (jQ) v += jQ[i * 64 + c];
    s_y[i * JLD + c] = v;
  }
  __syncthreads();
  if (stage != 4) {  // stage 4 = joint_SA_FFN alone on xq (the reference's Block module, CoevoDecoder.py:102-105)
    adaln_small8(s_y, s_a, s_gb + 0 * 128, J, tid);
    lin_tiled<64, 64>(s_a, JLD, w.wq, w.bq, s_q, JLD, J, tid, false, s_w, pre, w.proj_w, 64);
    __syncthreads();
    // ---- attention over the 431 vertex keys, 8 heads of 8: thread = (key slice s, query i, head h).  P = 8 J (query, head) pairs,
    // NSL = 512 / P key slices (3 at J = 17): slice s takes keys s, s + NSL, ... of every staged tile with its own online softmax
    // (m, l, o[8]); the slices' states are merged once at the end.  Four keys per step: one rescale per step, four independent dots.
    {
      const int P = J * 8, NSL = JS_THREADS / P;
      const int pr = tid % P, sl = tid / P;
      const int i = pr >> 3, h = pr & 7;
      const bool act = sl < NSL;
      float qv[8], o[8];
      const float scale = 0.35355339059327376220f * 1.44269504088896340736f;  // 8^-0.5 (8 heads of 8) * log2(e): 2^x softmax
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        qv[d] = act ? s_q[i * JLD + 8 * h + d] * scale : 0.f;
        o[d] = 0.f;
      }
      float m = -INFINITY, l = 0.f;
      const float* kvb = kv + (long long)b * NV * 128;
      // a tile of 64 keys (k | v: 32 KB) = 4 x 16 B per thread; the NEXT tile travels global -> registers under this tile's arithmetic
      f32x4 kvr[4];
      auto kv_fetch = [&](int j0) {
        const int nj = min(64, NV - j0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int idx = tid + q * JS_THREADS, r = idx >> 5, c4 = idx & 31;
          if (r < nj) kvr[q] = *reinterpret_cast<const f32x4*>(kvb + (long long)(j0 + r) * 128 + 4 * c4);
        }
      };
      kv_fetch(0);
      for (int j0 = 0; j0 < NV; j0 += 64) {
        const int nj = min(64, NV - j0);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int idx = tid + q * JS_THREADS, r = idx >> 5, c4 = idx & 31;
          if (r < nj) *reinterpret_cast<f32x4*>(&s_kv[r * 128 + 4 * c4]) = kvr[q];
        }
        __syncthreads();
        if (j0 + 64 < NV) kv_fetch(j0 + 64);
        if (act) {
          for (int r0 = sl; r0 < nj; r0 += 4 * NSL) {
            float sc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int r = r0 + u * NSL;
              if (r < nj) {
                const f32x4 k0 = *reinterpret_cast<const f32x4*>(&s_kv[r * 128 + 8 * h]);
                const f32x4 k1 = *reinterpret_cast<const f32x4*>(&s_kv[r * 128 + 8 * h + 4]);
                sc[u] = qv[0] * k0.x + qv[1] * k0.y + qv[2] * k0.z + qv[3] * k0.w + qv[4] * k1.x + qv[5] * k1.y + qv[6] * k1.z +
                        qv[7] * k1.w;
              } else {
                sc[u] = -INFINITY;
              }
            }
            const float mn = fmaxf(fmaxf(m, fmaxf(sc[0], sc[1])), fmaxf(sc[2], sc[3]));   // finite: key r0 exists
            const float corr = __builtin_amdgcn_exp2f(m - mn);
            l *= corr;
#pragma unroll
            for (int d = 0; d < 8; ++d) o[d] *= corr;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int r = r0 + u * NSL;
              if (r < nj) {
                const float pj = __builtin_amdgcn_exp2f(sc[u] - mn);
                l += pj;
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(&s_kv[r * 128 + 64 + 8 * h]);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(&s_kv[r * 128 + 64 + 8 * h + 4]);
                o[0] += pj * v0.x;
                o[1] += pj * v0.y;
                o[2] += pj * v0.z;
                o[3] += pj * v0.w;
                o[4] += pj * v1.x;
                o[5] += pj * v1.y;
                o[6] += pj * v1.z;
                o[7] += pj * v1.w;
              }
            }
            m = mn;
          }
        }
      }
      // merge the slices' states (s_h is free here): st[sl][pr] = {m, l, o[8]}
      if (act) {
        float* st = s_h + (sl * P + pr) * 10;
        st[0] = m;
        st[1] = l;
#pragma unroll
        for (int d = 0; d < 8; ++d) st[2 + d] = o[d];
      }
      __syncthreads();
      if (act && sl == 0) {
        float M = m;
        for (int s2 = 1; s2 < NSL; ++s2) M = fmaxf(M, s_h[(s2 * P + pr) * 10]);
        float L = 0.f, O[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int s2 = 0; s2 < NSL; ++s2) {
          const float* st = s_h + (s2 * P + pr) * 10;
          const float c = __builtin_amdgcn_exp2f(st[0] - M);   // (a slice that saw no key has m = -inf, l = 0: contributes 0)
          L += st[1] * c;
#pragma unroll
          for (int d = 0; d < 8; ++d) O[d] += st[2 + d] * c;
        }
        const float inv = 1.0f / L;
#pragma unroll
        for (int d = 0; d < 8; ++d) s_a[i * JLD + 8 * h + d] = O[d] * inv;
      }
    }
    lin_tiled<64, 64>(s_a, JLD, w.proj_w, w.proj_b, s_q, JLD, J, tid, false, s_w, pre, stage == 1 ? nullptr : w.fc1_w, 64);
    __syncthreads();
    for (int idx = tid; idx < J * 64; idx += JS_THREADS) {
      const int i = idx >> 6, c = idx & 63;
      s_y[i * JLD + c] += s_q[i * JLD + c];
    }
    __syncthreads();
    if (stage == 1) {
      for (int idx = tid; idx < J * 64; idx += JS_THREADS) y_out[(long long)b * J * 64 + idx] = s_y[(idx >> 6) * JLD + (idx & 63)];
      return;
    }
    // ---- FFN of the cross-attention block ----
    adaln_small8(s_y, s_a, s_gb + 1 * 128, J, tid);
    lin_tiled<64, 256>(s_a, JLD, w.fc1_w, w.fc1_b, s_h, 257, J, tid, true, s_w, pre, w.fc2_w, 256);
    lin_tiled<256, 64>(s_h, 257, w.fc2_w, w.fc2_b, s_q, JLD, J, tid, false, s_w, pre, stage == 2 ? nullptr : w.qkv_w, 64);
    __syncthreads();
    for (int idx = tid; idx < J * 64; idx += JS_THREADS) {
      const int i = idx >> 6, c = idx & 63;
      s_y[i * JLD + c] += s_q[i * JLD + c];
    }
    __syncthreads();
    if (stage == 2) {
      for (int idx = tid; idx < J * 64; idx += JS_THREADS) y_out[(long long)b * J * 64 + idx] = s_y[(idx >> 6) * JLD + (idx & 63)];
      return;
    }
  }
  // ---- self-attention block over the J joint tokens (8 heads of 8) ----
  adaln_small8(s_y, s_a, s_gb + 2 * 128, J, tid);
  lin_tiled<64, 192>(s_a, JLD, w.qkv_w, w.qkv_b, s_h, 193, J, tid, false, s_w, pre, w.sproj_w, 64);
  __syncthreads();
  {
    const int i = tid >> 3, h = tid & 7;
    if (i < J) {
      const float scale = 0.35355339059327376220f * 1.44269504088896340736f;
      float qv[8], o[8];
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        qv[d] = s_h[i * 193 + 8 * h + d] * scale;
        o[d] = 0.f;
      }
      float m = -INFINITY, l = 0.f;
      for (int j = 0; j < J; ++j) {
        float sc = 0.f;
#pragma unroll
        for (int d = 0; d < 8; ++d) sc += qv[d] * s_h[j * 193 + 64 + 8 * h + d];
        const float mn = fmaxf(m, sc);
        const float corr = __builtin_amdgcn_exp2f(m - mn), pj = __builtin_amdgcn_exp2f(sc - mn);
        l = l * corr + pj;
#pragma unroll
        for (int d = 0; d < 8; ++d) o[d] = o[d] * corr + pj * s_h[j * 193 + 128 + 8 * h + d];
        m = mn;
      }
      const float inv = 1.0f / l;
#pragma unroll
      for (int d = 0; d < 8; ++d) s_a[i * JLD + 8 * h + d] = o[d] * inv;
    }
  }
  lin_tiled<64, 64>(s_a, JLD, w.sproj_w, w.sproj_b, s_q, JLD, J, tid, false, s_w, pre, w.sfc1_w, 64);
  __syncthreads();
  for (int idx = tid; idx < J * 64; idx += JS_THREADS) {
    const int i = idx >> 6, c = idx & 63;
    s_y[i * JLD + c] += s_q[i * JLD + c];
  }
  __syncthreads();
  adaln_small8(s_y, s_a, s_gb + 3 * 128, J, tid);
  lin_tiled<64, 256>(s_a, JLD, w.sfc1_w, w.sfc1_b, s_h, 257, J, tid, true, s_w, pre, w.sfc2_w, 256);
  lin_tiled<256, 64>(s_h, 257, w.sfc2_w, w.sfc2_b, s_q, JLD, J, tid, false, s_w, pre, nullptr, 0);
  __syncthreads();
  for (int idx = tid; idx < J * 64; idx += JS_THREADS) {
    const int i = idx >> 6, c = idx & 63;
    s_y[i * JLD + c] += s_q[i * JLD + c];
  }
  __syncthreads();
  if (y_out)
    for (int idx = tid; idx < J * 64; idx += JS_THREADS) y_out[(long long)b * J * 64 + idx] = s_y[(idx >> 6) * JLD + (idx & 63)];
  // ---- proj_joint_feat2coor + residual on the ORIGINAL joints (CoevoDecoder.py:189) ----
  if (pose_out) {
    for (int idx = tid; idx < J * 3; idx += JS_THREADS) {
      const int i = idx / 3, k = idx % 3;
      float s = 0.f;
      for (int c = 0; c < 64; ++c) s += s_cw[k * 64 + c] * s_y[i * JLD + c];
      pose_out[((long long)b * J + i) * 3 + k] = (s + s_cw[192 + k]) + jt[((long long)b * J + i) * 3 + k];
    }
  }
}

// ======================================================================================================
// tail: build the packed operand of the final GEMM and the J_regressor projection
// ======================================================================================================
// A'[b][0:2048] = relu(g[b]) ; A'[b][2048 + 3v + l] = vt[b][v][l] ; A'[b][3341:KP] = 0   (CoevoDecoder.py:238-244)
// PACKED: the row is written pre-split ([KP/16][16 f16 hi | 16 f16 lo*2^11] in the bytes of the fp32 row, the layout the lifter's producers
// write): the final product then spends no vector instruction on splitting its A operand (the split is the same two roundings either way:
// bit-identical results).
template <bool PACKED>
__global__ __launch_bounds__(256) void build_final_operand_kernel(const float* __restrict__ g, const float* __restrict__ vt,
                                                                  float* __restrict__ A, int B, int KP) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)B * KP) return;
  const int b = (int)(idx / KP), k = (int)(idx % KP);
  float v = 0.f;
  if (k < 2048)
    v = fmaxf(g[(long long)b * 2048 + k], 0.f);
  else if (k < 2048 + NV * 3)
    v = vt[(long long)b * NV * 3 + (k - 2048)];
  if constexpr (PACKED) {
    _Float16* row = reinterpret_cast<_Float16*>(A + (long long)b * KP);
    const _Float16 h = (_Float16)v;
    row[(k >> 4) * 32 + (k & 15)] = h;
    row[(k >> 4) * 32 + 16 + (k & 15)] = (_Float16)((v - (float)h) * 2048.0f);
  } else {
    A[idx] = v;
  }
}

// joints_mm[b][j][l] = sum_nz data * (mesh[b][col][l] * 1000)   (lib/core/base.py:223-225), CSR regressor
__global__ __launch_bounds__(256) void j_regress_kernel(const float* __restrict__ mesh, const int* __restrict__ indptr,
                                                        const int* __restrict__ indices, const float* __restrict__ data,
                                                        float* __restrict__ out, int B, int R, int NVF, float scale) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * R * 3) return;
  const int l = idx % 3, j = (idx / 3) % R, b = idx / (3 * R);
  const float* m = mesh + (long long)b * NVF * 3;
  float s = 0.f;
  for (int e = indptr[j]; e < indptr[j + 1]; ++e) s += data[e] * (m[(long long)indices[e] * 3 + l] * scale);
  out[idx] = s;
}

// ======================================================================================================
// C-ABI launchers
// ======================================================================================================
// waves per workgroup of the two FFN-carrying kernels (two per SIMD: one wave per SIMD measured 33 % slower, profiles/r05_b_*)
#define MLP_WAVES 8
#define CAM_WAVES 7
static int mlp_grid(int B) {
  const int tiles = B * NTILE;
  int g = (tiles + MLP_WAVES - 1) / MLP_WAVES;
  return g < 256 ? g : 256;
}
static int tl_grid(int B, int per_cu) {
  const int tiles = B * NTILE;
  int g = (tiles + 3) / 4;
  const int cap = 256 * per_cu;
  return g < cap ? g : cap;
}

extern "C" int pmce_vertex_init_gather_f32(const float* joints, const int* vj, float* vt, int B, int J, hipStream_t stream) {
  PMCE_REQUIRE(joints && vj && vt && B > 0 && J > 0, "vertex_init_gather: bad args");
  const int n = B * NV * 3;
  hipLaunchKernelGGL(vertex_init_gather_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, joints, vj, vt, B, J);
  return pmce_check_launch("vertex_init_gather");
}

static int launch_ca_fold(const CaFoldArgs& a, int B, hipStream_t stream, const char* what) {
  PMCE_REQUIRE(a.J >= 1 && a.J <= 32 && B > 0, "%s: J must be in 1..32", what);
  PMCE_REQUIRE(!a.img || (a.J <= 23 && (reinterpret_cast<uintptr_t>(a.img) & 15) == 0), "%s: the f16 image needs J <= 23 and a 16-byte aligned buffer", what);
  hipLaunchKernelGGL(ca_fold_kernel, dim3(B), dim3(JS_THREADS), 0, stream, a);
  return pmce_check_launch(what);
}
extern "C" int pmce_ca_image_floats(void) { return CA_IMG_STRIDE; }
// img (may be null): the operands' f16 image per clip for pmce_vertex_ca_mlp_pk_f32's split_f16 form, [B][pmce_ca_image_floats()], J <= 23.
// Kf / s0 / Vf may each be null when only the image is wanted.
extern "C" int pmce_ca_fold_img_f32(const float* xk, const float* xv, const float* GB, int gb_stride, int iq, int ik, int iv,
                                    const float* Wq, const float* bq, const float* Wk, const float* bk, const float* Wv,
                                    const float* bv, const float* Wp, float* Kf, float* s0, float* Vf, float* img, int B, int J,
                                    hipStream_t stream) {
  PMCE_REQUIRE(xk && xv && GB && Wq && bq && Wk && bk && Wv && bv && Wp, "ca_fold: null pointer");
  CaFoldArgs a{};
  a.xk = xk; a.xv = xv; a.GB = GB; a.gb_stride = gb_stride; a.iq = iq; a.ik = ik; a.iv = iv;
  a.Wq = Wq; a.bq = bq; a.Wk = Wk; a.bk = bk; a.Wv = Wv; a.bv = bv; a.Wp = Wp;
  a.Kf = Kf; a.s0 = s0; a.Vf = Vf; a.img = img; a.J = J;
  return launch_ca_fold(a, B, stream, "ca_fold");
}
extern "C" int pmce_ca_fold_f32(const float* xk, const float* xv, const float* GB, int gb_stride, int iq, int ik, int iv,
                                const float* Wq, const float* bq, const float* Wk, const float* bk, const float* Wv,
                                const float* bv, const float* Wp, float* Kf, float* s0, float* Vf, int B, int J,
                                hipStream_t stream) {
  return pmce_ca_fold_img_f32(xk, xv, GB, gb_stride, iq, ik, iv, Wq, bq, Wk, bk, Wv, bv, Wp, Kf, s0, Vf, nullptr, B, J, stream);
}
// The joint side of a CoevoBlock in one launch: the joint embedding (CoevoDecoder.py:177-180,184: jf = joint_proj(jt) + joint_pos_embed,
// xk = proj_j2v_dim(jf) + j2v_K_embed, xv = jf) followed by the fold.  jf_out [B,J,64] may be null (only the joint stream reads it).
extern "C" int pmce_joint_prep_f32(const float* jt, const float* Wj, const float* bj, const float* jpos, const float* Wj2v,
                                   const float* bj2v, const float* j2vK, float* jf_out, const float* GB, int gb_stride, int iq, int ik,
                                   int iv, const float* Wq, const float* bq, const float* Wk, const float* bk, const float* Wv,
                                   const float* bv, const float* Wp, float* Kf, float* s0, float* Vf, float* img, int B, int J,
                                   hipStream_t stream) {
  PMCE_REQUIRE(jt && Wj && bj && jpos && Wj2v && bj2v && j2vK && GB && Wq && bq && Wk && bk && Wv && bv && Wp, "joint_prep: null pointer");
  CaFoldArgs a{};
  a.jt = jt; a.Wj = Wj; a.bj = bj; a.jpos = jpos; a.Wj2v = Wj2v; a.bj2v = bj2v; a.j2vK = j2vK; a.jf_out = jf_out;
  a.GB = GB; a.gb_stride = gb_stride; a.iq = iq; a.ik = ik; a.iv = iv;
  a.Wq = Wq; a.bq = bq; a.Wk = Wk; a.bk = bk; a.Wv = Wv; a.bv = bv; a.Wp = Wp;
  a.Kf = Kf; a.s0 = s0; a.Vf = Vf; a.img = img; a.J = J;
  return launch_ca_fold(a, B, stream, "joint_prep");
}

extern "C" int pmce_vertex_ca_f32(const float* xq, const float* vt, const float* Wv3, const float* Eq, const float* Kf,
                                  const float* s0, const float* Vf, const float* bp, float* out, int B, int J,
                                  hipStream_t stream) {
  PMCE_REQUIRE((xq || (vt && Wv3 && Eq)) && Kf && s0 && Vf && bp && out, "vertex_ca: null pointer");
  PMCE_REQUIRE(J >= 1 && J <= 32 && B > 0, "vertex_ca: J must be in 1..32");
  hipLaunchKernelGGL(vertex_ca_kernel, dim3(2, B), dim3(448), 0, stream, xq, vt, Wv3, Eq, Kf, s0, Vf, bp, out, J);
  return pmce_check_launch("vertex_ca");
}

// ffn_img: the FFN's pre-made LDS image of the f16 form (pmce_ffn_pack_f16; null = made by every workgroup from W1 / W2)
extern "C" int pmce_adaln_mlp_pk_f32(const float* xin, const float* GB, int gb_stride, int inst, const float* W1,
                                     const float* b1, const float* W2, const float* b2, float* yout, const float* Wc,
                                     const float* bc, const float* vt_in, float* vt_out, int B, int split_f16,
                                     const float* ffn_img, hipStream_t stream) {
  PMCE_REQUIRE(xin && GB && W1 && b1 && W2 && b2 && (yout || vt_out), "adaln_mlp: null pointer");
  PMCE_REQUIRE(!vt_out || (Wc && bc && vt_in), "adaln_mlp: coordinate head needs Wc, bc, vt_in");
  const size_t lds = (size_t)(256 * LDW64 + 64 * LDW256 + 256 + 64 + 32 + 256) * sizeof(float);
  static std::atomic<unsigned long long> attr{0}, attr_s{0};
  if (split_f16) {
    PMCE_TRY(pmce_opt_in_lds((const void*)adaln_mlp_kernel<true, MLP_WAVES>, (int)lds, attr_s, "adaln_mlp"));
    hipLaunchKernelGGL((adaln_mlp_kernel<true, MLP_WAVES>), dim3(mlp_grid(B)), dim3(64 * MLP_WAVES), lds, stream, xin, GB, gb_stride, inst, W1, b1, W2, b2,
                       yout, Wc, bc, vt_in, vt_out, B, ffn_img);
  } else {
    PMCE_TRY(pmce_opt_in_lds((const void*)adaln_mlp_kernel<false, MLP_WAVES>, (int)lds, attr, "adaln_mlp"));
    hipLaunchKernelGGL((adaln_mlp_kernel<false, MLP_WAVES>), dim3(mlp_grid(B)), dim3(64 * MLP_WAVES), lds, stream, xin, GB, gb_stride, inst, W1, b1, W2, b2,
                       yout, Wc, bc, vt_in, vt_out, B, nullptr);
  }
  return pmce_check_launch("adaln_mlp");
}
extern "C" int pmce_adaln_mlp_f32(const float* xin, const float* GB, int gb_stride, int inst, const float* W1,
                                  const float* b1, const float* W2, const float* b2, float* yout, const float* Wc,
                                  const float* bc, const float* vt_in, float* vt_out, int B, hipStream_t stream) {
  return pmce_adaln_mlp_pk_f32(xin, GB, gb_stride, inst, W1, b1, W2, b2, yout, Wc, bc, vt_in, vt_out, B, 0, nullptr, stream);
}

// split_f16 != 0: the FFN AND the cross-attention in the three-product f16 form - ffn_img (pmce_ffn_pack_f16; null = converted per
// workgroup from W1 / W2) and ca_img (required: pmce_ca_fold_img_f32's / pmce_joint_prep_f32's image of the folded operands; Kf / s0 /
// Vf are then not read and may be null).  J > 23: the two-launch fp32-attention form through `scratch` (needs Kf / s0 / Vf).
extern "C" int pmce_vertex_ca_mlp_pk_f32(const float* xq, const float* vt, const float* Wv3, const float* Eq, const float* Kf,
                                         const float* s0, const float* Vf, const float* bp, const float* GB, int gb_stride,
                                         int inst, const float* W1, const float* b1, const float* W2, const float* b2,
                                         float* yout, float* scratch, int B, int J, int split_f16, const float* ffn_img,
                                         const float* ca_img, hipStream_t stream) {
  PMCE_REQUIRE((xq || (vt && Wv3 && Eq)) && bp && GB && W1 && b1 && W2 && b2 && yout, "vertex_ca_mlp: null pointer");
  PMCE_REQUIRE(J >= 1 && J <= 32 && B > 0, "vertex_ca_mlp: J must be in 1..32");
  if (J > 23) {  // one clip's folded operands no longer fit beside the FFN weights: the two-launch form
    PMCE_REQUIRE(scratch && Kf && s0 && Vf, "vertex_ca_mlp: J > 23 needs Kf / s0 / Vf and a [B,431,64] scratch buffer");
    PMCE_TRY(pmce_vertex_ca_f32(xq, vt, Wv3, Eq, Kf, s0, Vf, bp, scratch, B, J, stream));
    return pmce_adaln_mlp_pk_f32(scratch, GB, gb_stride, inst, W1, b1, W2, b2, yout, nullptr, nullptr, nullptr, nullptr, B, split_f16,
                                 ffn_img, stream);
  }
  const size_t lds = (size_t)(256 * LDW64 + 64 * LDW256 + 256 + 64 + 32 + CA_IMG_FLOATS(J)) * sizeof(float);
  static std::atomic<unsigned long long> attr{0}, attr_s{0};
  const int g = 2 * B < 256 ? 2 * B : 256;
  if (split_f16) {
    PMCE_REQUIRE(ca_img && (reinterpret_cast<uintptr_t>(ca_img) & 15) == 0, "vertex_ca_mlp: the split_f16 form needs the operands' image (pmce_ca_fold_img_f32)");
    PMCE_TRY(pmce_opt_in_lds((const void*)vertex_ca_mlp_kernel<true, CAM_WAVES>, 163840, attr_s, "vertex_ca_mlp"));
    hipLaunchKernelGGL((vertex_ca_mlp_kernel<true, CAM_WAVES>), dim3(g), dim3(64 * CAM_WAVES), lds, stream, xq, vt, Wv3, Eq, Kf, s0, Vf, bp, GB, gb_stride,
                       inst, W1, b1, W2, b2, yout, B, J, ffn_img, ca_img);
  } else {
    PMCE_REQUIRE(Kf && s0 && Vf, "vertex_ca_mlp: null pointer");
    PMCE_TRY(pmce_opt_in_lds((const void*)vertex_ca_mlp_kernel<false, CAM_WAVES>, 163840, attr, "vertex_ca_mlp"));
    hipLaunchKernelGGL((vertex_ca_mlp_kernel<false, CAM_WAVES>), dim3(g), dim3(64 * CAM_WAVES), lds, stream, xq, vt, Wv3, Eq, Kf, s0, Vf, bp, GB, gb_stride,
                       inst, W1, b1, W2, b2, yout, B, J, nullptr, nullptr);
  }
  return pmce_check_launch("vertex_ca_mlp");
}
extern "C" int pmce_vertex_ca_mlp_f32(const float* xq, const float* vt, const float* Wv3, const float* Eq, const float* Kf,
                                      const float* s0, const float* Vf, const float* bp, const float* GB, int gb_stride,
                                      int inst, const float* W1, const float* b1, const float* W2, const float* b2, float* yout,
                                      float* scratch, int B, int J, hipStream_t stream) {
  return pmce_vertex_ca_mlp_pk_f32(xq, vt, Wv3, Eq, Kf, s0, Vf, bp, GB, gb_stride, inst, W1, b1, W2, b2, yout, scratch, B, J, 0, nullptr,
                                   nullptr, stream);
}

extern "C" int pmce_adaln_qkv_f32(const float* xin, const float* GB, int gb_stride, int inst, const float* Wqkv,
                                  const float* bqkv, float* qkv, int B, hipStream_t stream) {
  PMCE_REQUIRE(xin && GB && Wqkv && bqkv && qkv && B > 0, "adaln_qkv: null pointer");
  hipLaunchKernelGGL(adaln_qkv_kernel, dim3(tl_grid(B, 2)), dim3(256), 0, stream, xin, GB, gb_stride, inst, Wqkv, bqkv, qkv, B);
  return pmce_check_launch("adaln_qkv");
}
// The qkv weight's f16 image for pmce_vertex_sab_split_f32: made once per weight.
__global__ __launch_bounds__(256) void qkv_pack_kernel(const float* __restrict__ W, float* __restrict__ img) {
  __shared__ float red[16];
  float up, down;
  weight_scale(W, 192 * 64, red, threadIdx.x, 256, up, down);
  stage_weight_split<64>(img, W, 192, up, threadIdx.x, 256);
  if (threadIdx.x == 0) {
    img[192 * LDW64] = up;
    img[192 * LDW64 + 1] = down;
  }
}
extern "C" int pmce_qkv_image_floats(void) { return QKV_IMG_FLOATS; }
extern "C" int pmce_qkv_pack_f16(const float* Wqkv, float* img, hipStream_t stream) {
  PMCE_REQUIRE(Wqkv && img && (reinterpret_cast<uintptr_t>(img) & 15) == 0, "qkv_pack: null or unaligned pointer");
  hipLaunchKernelGGL(qkv_pack_kernel, dim3(1), dim3(256), 0, stream, Wqkv, img);
  return pmce_check_launch("qkv_pack_f16");
}

extern "C" int pmce_vertex_sa_f32(const float* xin, const float* qkv, const float* Wp, const float* bp, float* yout, int B,
                                  hipStream_t stream) {
  PMCE_REQUIRE(xin && qkv && Wp && bp && yout && B > 0, "vertex_sa: null pointer");
  hipLaunchKernelGGL(vertex_sa_kernel, dim3(2, B), dim3(448), 0, stream, xin, qkv, Wp, bp, yout);
  return pmce_check_launch("vertex_sa");
}

// The attention half of the vertex stream's AdaLN Block in one launch (split mode): AdaLN + qkv product + 431 x 431 attention + proj +
// residual.  scratch: pmce_vertex_sab_scratch_floats(B) floats, 16-byte aligned (the clip's key tiles as f16 planes; contents are
// meaningless outside the call).  qkv_img: pmce_qkv_pack_f16(Wqkv).
// Non-decreasing in B (two tile sets per clip up to B = 128, one beyond - never less than the 256 sets of B = 128): a caller that sizes
// once for its largest batch holds enough for every smaller one.
extern "C" long long pmce_vertex_sab_scratch_floats(int B) {
  const long long sets = B > 128 ? (B > 256 ? B : 256) : 2LL * B;
  return sets * NTILE * SAB_TILE_FLOATS;
}
extern "C" int pmce_vertex_sab_split_f32(const float* xin, const float* GB, int gb_stride, int inst, const float* qkv_img,
                                         const float* bqkv, const float* Wp, const float* bp, float* scratch, float* yout, int B,
                                         hipStream_t stream) {
  PMCE_REQUIRE(xin && GB && qkv_img && bqkv && Wp && bp && scratch && yout && B > 0, "vertex_sab: null pointer");
  PMCE_REQUIRE((reinterpret_cast<uintptr_t>(scratch) & 15) == 0 && (reinterpret_cast<uintptr_t>(qkv_img) & 15) == 0, "vertex_sab: unaligned pointer");
  if (B > 128)
    hipLaunchKernelGGL(vertex_sab_kernel<2>, dim3(1, B), dim3(448), 0, stream, xin, GB, gb_stride, inst, qkv_img, bqkv, Wp, bp, scratch, yout);
  else
    hipLaunchKernelGGL(vertex_sab_kernel<1>, dim3(2, B), dim3(448), 0, stream, xin, GB, gb_stride, inst, qkv_img, bqkv, Wp, bp, scratch, yout);
  return pmce_check_launch("vertex_sab");
}

extern "C" int pmce_tkv_image_floats(void) { return TKV_IMG_FLOATS; }
extern "C" int pmce_tkv_pack_f16(const float* Wv2j, const float* Wk, const float* Wv, float* img, hipStream_t stream) {
  PMCE_REQUIRE(Wv2j && Wk && Wv && img && (reinterpret_cast<uintptr_t>(img) & 15) == 0, "tkv_pack: null or unaligned pointer");
  hipLaunchKernelGGL(tkv_pack_kernel, dim3(1), dim3(256), 0, stream, Wv2j, Wk, Wv, img);
  return pmce_check_launch("tkv_pack_f16");
}
// tkv_img != null: the three 64 x 64 products in the three-product f16 form from the image of pmce_tkv_pack_f16(Wv2j, Wk, Wv) (Wv2j, Wk, Wv
// themselves are then not read)
extern "C" int pmce_tokens_kv_pk_f32(const float* xk, const float* xv, const float* vt, const float* Wv3, const float* Ev,
                                     const float* Wv2j, const float* Ek, const float* GB, int gb_stride, int ik, int iv,
                                     const float* Wk, const float* bk, const float* Wv, const float* bv, float* kv, int B,
                                     const float* tkv_img, hipStream_t stream) {
  PMCE_REQUIRE(((xk && xv) || (vt && Wv3 && Ev && Ek && (Wv2j || tkv_img))) && GB && bk && bv && kv && ((Wk && Wv) || tkv_img),
               "tokens_kv: null pointer");
  if (tkv_img) {
    PMCE_REQUIRE((reinterpret_cast<uintptr_t>(tkv_img) & 15) == 0, "tokens_kv: unaligned image");
    hipLaunchKernelGGL(tokens_kv_kernel<true>, dim3(tl_grid(B, 2)), dim3(256), 0, stream, xk, xv, vt, Wv3, Ev, Wv2j, Ek, GB, gb_stride, ik, iv,
                       Wk, bk, Wv, bv, kv, B, tkv_img);
  } else {
    hipLaunchKernelGGL(tokens_kv_kernel<false>, dim3(tl_grid(B, 2)), dim3(256), 0, stream, xk, xv, vt, Wv3, Ev, Wv2j, Ek, GB, gb_stride, ik, iv,
                       Wk, bk, Wv, bv, kv, B, nullptr);
  }
  return pmce_check_launch("tokens_kv");
}
extern "C" int pmce_tokens_kv_f32(const float* xk, const float* xv, const float* vt, const float* Wv3, const float* Ev,
                                  const float* Wv2j, const float* Ek, const float* GB, int gb_stride, int ik, int iv,
                                  const float* Wk, const float* bk, const float* Wv, const float* bv, float* kv, int B,
                                  hipStream_t stream) {
  return pmce_tokens_kv_pk_f32(xk, xv, vt, Wv3, Ev, Wv2j, Ek, GB, gb_stride, ik, iv, Wk, bk, Wv, bv, kv, B, nullptr, stream);
}

extern "C" int pmce_joint_stream_f32(const float* xq, const float* jQ, const float* kv, const float* GB, int gb_stride,
                                     const float* const* wptr, const int* inst, const float* jt, float* y_out,
                                     float* pose_out, int B, int J, int stage, hipStream_t stream) {
  PMCE_REQUIRE(xq && (kv || stage == 4) && GB && wptr && inst, "joint_stream: null pointer");
  PMCE_REQUIRE(J >= 1 && J <= 32 && B > 0, "joint_stream: J must be in 1..32");
  PMCE_REQUIRE(stage >= 1 && stage <= 4, "joint_stream: stage must be 1 (CA), 2 (CA+FFN), 3 (full) or 4 (SA block only)");
  JointStreamW w;
  w.wq = wptr[0]; w.bq = wptr[1]; w.proj_w = wptr[2]; w.proj_b = wptr[3];
  w.fc1_w = wptr[4]; w.fc1_b = wptr[5]; w.fc2_w = wptr[6]; w.fc2_b = wptr[7];
  w.qkv_w = wptr[8]; w.qkv_b = wptr[9]; w.sproj_w = wptr[10]; w.sproj_b = wptr[11];
  w.sfc1_w = wptr[12]; w.sfc1_b = wptr[13]; w.sfc2_w = wptr[14]; w.sfc2_b = wptr[15];
  w.coor_w = wptr[16]; w.coor_b = wptr[17];
  w.i_normq = inst[0]; w.i_norm2 = inst[1]; w.i_snorm1 = inst[2]; w.i_snorm2 = inst[3];
  hipLaunchKernelGGL(joint_stream_kernel, dim3(B), dim3(JS_THREADS), 0, stream, xq, jQ, kv, GB, gb_stride, w, jt, y_out, pose_out, J,
                     stage);
  return pmce_check_launch("joint_stream");
}

extern "C" int pmce_build_final_operand_pk_f32(const float* g, const float* vt, float* A, int B, int KP, int packed, hipStream_t stream) {
  PMCE_REQUIRE(g && vt && A && KP >= 2048 + NV * 3 && (!packed || KP % 16 == 0), "build_final_operand: bad args");
  const long long n = (long long)B * KP;
  if (packed) hipLaunchKernelGGL(build_final_operand_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, g, vt, A, B, KP);
  else hipLaunchKernelGGL(build_final_operand_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, g, vt, A, B, KP);
  return pmce_check_launch("build_final_operand");
}
extern "C" int pmce_build_final_operand_f32(const float* g, const float* vt, float* A, int B, int KP, hipStream_t stream) {
  return pmce_build_final_operand_pk_f32(g, vt, A, B, KP, 0, stream);
}

extern "C" int pmce_j_regress_f32(const float* mesh, const int* indptr, const int* indices, const float* data, float* out,
                                  int B, int R, int NVF, float scale, hipStream_t stream) {
  PMCE_REQUIRE(mesh && indptr && indices && data && out && B > 0 && R > 0, "j_regress: bad args");
  const int n = B * R * 3;
  hipLaunchKernelGGL(j_regress_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, mesh, indptr, indices, data, out, B, R, NVF,
                     scale);
  return pmce_check_launch("j_regress");
}
