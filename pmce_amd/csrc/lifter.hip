// Temporal pose encoder ("lifter") kernels other than the Linear layers (those are gemm_f32.hip).
// Reference: lib/models/PoseEstimation.py (GraphormerNet) + timm Attention.
//
// Activations keep ONE layout for the whole encoder: x[b][t][j][c] (token = (b*T+t)*J + j).  The reference's
// '(b t) j c <-> (b j) t c' rearranges (PoseEstimation.py:87,101,104,109; 32 % of its CPU time) never
// happen: only the attention kernel needs to know which tokens form a sequence, and it gets that as strides.
#include "common.hpp"

// ------------------------------------------------------------------------------------------------------
// token embedding  (PoseEstimation.py:78-81)
//   x[tok][c] = Wje[c][0]*p0 + Wje[c][1]*p1 + bje[c] + E[b*T+t][c] + spos[j][c]
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_tokens_kernel(const float* __restrict__ pose2d, const float* __restrict__ E,
                                                           const float* __restrict__ Wje, const float* __restrict__ bje,
                                                           const float* __restrict__ spos, float* __restrict__ x,
                                                           long long ntok, int J, int C) {
  const int c4n = C >> 2;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= ntok * c4n) return;
  const long long tok = idx / c4n;
  const int c = (int)(idx % c4n) * 4;
  const long long bt = tok / J;
  const int j = (int)(tok % J);
  const float p0 = pose2d[tok * 2 + 0], p1 = pose2d[tok * 2 + 1];
  const f32x4 e = *reinterpret_cast<const f32x4*>(E + bt * C + c);
  const f32x4 sp = *reinterpret_cast<const f32x4*>(spos + (long long)j * C + c);
  const f32x4 bj = *reinterpret_cast<const f32x4*>(bje + c);
  f32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    // same association as the reference: (joint_embed(x)) + imgfeat_embed + spatial_pos
    const float je = Wje[(c + i) * 2 + 0] * p0 + Wje[(c + i) * 2 + 1] * p1 + bj[i];
    o[i] = (je + e[i]) + sp[i];
  }
  *reinterpret_cast<f32x4*>(x + tok * C + c) = o;
}

// ------------------------------------------------------------------------------------------------------
// LayerNorm chain, one wavefront per token row:
//   y1 = w1 ? LN(x; w1,b1,eps1) : x ;  y1 += add[(row / add_div) % add_mod]   (temporal_pos_embed, :88)
//   out1 = y1 (if out1) ;  out2 = LN(y1; w2,b2,eps2) (if out2)
// covers norm_s/norm_t (shared across depth, eps 1e-6, :38,84,92) fused with the NEXT block's norm1/norm2.
// ------------------------------------------------------------------------------------------------------
template <int C>
__device__ __forceinline__ void ln_regs(float* v, const float* __restrict__ w, const float* __restrict__ b, float eps,
                                        int lane) {
  constexpr int NV = C / 64;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) s += v[i];
  const float mean = wave_sum(s) * (1.0f / C);
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float d = v[i] - mean;
    ss += d * d;
  }
  const float var = wave_sum(ss) * (1.0f / C);
  const float inv = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int i4 = 0; i4 < NV / 4; ++i4) {
    const int c = i4 * 256 + lane * 4;
    const f32x4 wv = *reinterpret_cast<const f32x4*>(w + c);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(b + c);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i4 * 4 + i] = (v[i4 * 4 + i] - mean) * inv * wv[i] + bv[i];
  }
}

// The same with the weight and bias of this lane's channels already in registers (kernels that normalise several rows per wavefront: fetched
// per row, the parameters are more vector-cache traffic than the row itself).  The same operations in the same order: the same bits.
template <int C>
__device__ __forceinline__ void ln_regs_pre(float* v, const float* wreg, const float* breg, float eps) {
  constexpr int NV = C / 64;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) s += v[i];
  const float mean = wave_sum(s) * (1.0f / C);
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float d = v[i] - mean;
    ss += d * d;
  }
  const float var = wave_sum(ss) * (1.0f / C);
  const float inv = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = (v[i] - mean) * inv * wreg[i] + breg[i];
}
// this lane's channels (256 i4 + 4 lane + [0, 4)) of a per-channel vector
template <int C>
__device__ __forceinline__ void lane_channels(const float* __restrict__ p, int lane, float* out) {
#pragma unroll
  for (int i4 = 0; i4 < C / 256; ++i4) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p + i4 * 256 + lane * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i4 * 4 + i] = t[i];
  }
}

// Optional window gather (streaming): with win != nullptr, output row (w*T + t)*J + j reads input row
// frame(w,t)*J + j, frame = win[2w] + t (or win[2w] when win[2w] == win[2w+1]: a single frame repeated).
template <int C>
__global__ __launch_bounds__(256) void ln_chain_kernel(const float* __restrict__ x, long long rows,
                                                       const float* __restrict__ w1, const float* __restrict__ b1,
                                                       float eps1, const float* __restrict__ add, int add_div, int add_mod,
                                                       float* __restrict__ out1, const float* __restrict__ w2,
                                                       const float* __restrict__ b2, float eps2, float* __restrict__ out2,
                                                       const int* __restrict__ win, int nframes, int out2_split) {
  constexpr int NV = C / 64;
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  long long src = row;
  if (win) {  // add_div = J, add_mod = T in this mode
    const long long wt = row / add_div;
    const int j = (int)(row % add_div), t = (int)(wt % add_mod);
    const long long w = wt / add_mod;
    const int s0 = win[2 * w], e0 = win[2 * w + 1];
    const int fr = min(max(s0 == e0 ? s0 : s0 + t, 0), nframes - 1);
    src = (long long)fr * add_div + j;
  }
  float v[NV];
#pragma unroll
  for (int i4 = 0; i4 < NV / 4; ++i4) {
    // Streaming (non-temporal) accesses for the row itself and for the fp32 residual stream out1 (round 4): read once / next touched a
    // product later.  NOT for out2: it is the NEXT product's A operand, and written back normally that product finds it in the Infinity
    // Cache (profiles/r04_g_ln_chain_nontemporal_ab.txt: ln_chain -13 %, the attention after it -5 %, the lifter products -1.6 %).
    const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x + src * C + i4 * 256 + lane * 4));
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i4 * 4 + i] = t[i];
  }
  if (w1) ln_regs<C>(v, w1, b1, eps1, lane);
  if (add) {
    const float* a = add + (long long)((row / add_div) % add_mod) * C;
#pragma unroll
    for (int i4 = 0; i4 < NV / 4; ++i4) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(a + i4 * 256 + lane * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i4 * 4 + i] += t[i];
    }
  }
  if (out1) {
#pragma unroll
    for (int i4 = 0; i4 < NV / 4; ++i4) {
      f32x4 t;
#pragma unroll
      for (int i = 0; i < 4; ++i) t[i] = v[i4 * 4 + i];
      __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(out1 + row * C + i4 * 256 + lane * 4));
    }
  }
  if (out2) {
    ln_regs<C>(v, w2, b2, eps2, lane);
#pragma unroll
    for (int i4 = 0; i4 < NV / 4; ++i4) {
      f32x4 t;
#pragma unroll
      for (int i = 0; i < 4; ++i) t[i] = v[i4 * 4 + i];
      if (out2_split) store4_split_f16(out2 + row * C, i4 * 256 + lane * 4, t);  // operand of a three-product f16 GEMM
      else *reinterpret_cast<f32x4*>(out2 + row * C + i4 * 256 + lane * 4) = t;
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// Token embedding + the first LayerNorm in one pass (round 6): embed_tokens followed by ln_chain(out2 = norm1 of SpatialBlocks[0]) wrote the
// tokens and read them straight back (143 MB at B = 256, C = 512).  One wavefront per token computes the row in registers - lane l holds
// channels 256 i4 + 4 l + [0, 4) as ln_chain does -, stores it (x: the residual stream) and stores its LayerNorm (xn: the first product's A
// operand, pre-split when xn_split).  The same operations in the same order as the two kernels: bit-identical.
// ------------------------------------------------------------------------------------------------------
// A frame's J tokens are shared by `wpf` wavefronts (wavefront w of the frame takes j = w, w + wpf, ...): the frame's image-feature row, the
// embedding's per-channel constants and the LayerNorm's parameters are fetched once per WAVEFRONT and serve all its tokens.  Fetched per token
// (one token per wavefront, as until round 6) they are 14 KB through the vector cache for 4 KB of stores - that, not HBM, bounded the kernel
// (88 us for 285 MB of stores at B = 256, C = 512).  The launcher picks wpf so that there are enough wavefronts to fill the chip: 2 at
// B = 256, J (one token per wavefront, the old form) for a single clip.
template <int C>
__global__ __launch_bounds__(256) void embed_ln_kernel(const float* __restrict__ pose2d, const float* __restrict__ E,
                                                       const float* __restrict__ Wje, const float* __restrict__ bje,
                                                       const float* __restrict__ spos, float* __restrict__ x, long long nframes, int J,
                                                       const float* __restrict__ w2, const float* __restrict__ b2, float eps2,
                                                       float* __restrict__ xn, int xn_split, int wpf) {
  constexpr int NV = C / 64;
  const int lane = threadIdx.x & 63;
  const unsigned wid = blockIdx.x * 4u + (threadIdx.x >> 6);  // (ntok < 2^31: 32-bit division)
  const long long bt = wid / (unsigned)wpf;
  if (bt >= nframes) return;
  float e[NV], bj[NV], wa[NV], wb[NV], lw[NV], lb[NV];
  lane_channels<C>(E + bt * C, lane, e);
  lane_channels<C>(bje, lane, bj);
  lane_channels<C>(w2, lane, lw);
  lane_channels<C>(b2, lane, lb);
#pragma unroll
  for (int i4 = 0; i4 < NV / 4; ++i4) {
    const int c = i4 * 256 + lane * 4;
    const f32x4 w01 = *reinterpret_cast<const f32x4*>(Wje + c * 2), w23 = *reinterpret_cast<const f32x4*>(Wje + c * 2 + 4);
    wa[i4 * 4 + 0] = w01[0]; wa[i4 * 4 + 1] = w01[2]; wa[i4 * 4 + 2] = w23[0]; wa[i4 * 4 + 3] = w23[2];
    wb[i4 * 4 + 0] = w01[1]; wb[i4 * 4 + 1] = w01[3]; wb[i4 * 4 + 2] = w23[1]; wb[i4 * 4 + 3] = w23[3];
  }
  for (int j = (int)(wid % (unsigned)wpf); j < J; j += wpf) {
    const long long tok = bt * J + j;
    const float p0 = pose2d[tok * 2 + 0], p1 = pose2d[tok * 2 + 1];
    float v[NV];
#pragma unroll
    for (int i4 = 0; i4 < NV / 4; ++i4) {
      const int c = i4 * 256 + lane * 4;
      const f32x4 sp = *reinterpret_cast<const f32x4*>(spos + (long long)j * C + c);
      f32x4 o;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // same association as the reference (and as embed_tokens_kernel): (joint_embed(x)) + imgfeat_embed + spatial_pos
        const float je = wa[i4 * 4 + i] * p0 + wb[i4 * 4 + i] * p1 + bj[i4 * 4 + i];
        o[i] = (je + e[i4 * 4 + i]) + sp[i];
        v[i4 * 4 + i] = o[i];
      }
      __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(x + tok * C + c));
    }
    ln_regs_pre<C>(v, lw, lb, eps2);
#pragma unroll
    for (int i4 = 0; i4 < NV / 4; ++i4) {
      f32x4 t;
#pragma unroll
      for (int i = 0; i < 4; ++i) t[i] = v[i4 * 4 + i];
      if (xn_split) store4_split_f16(xn + tok * C, i4 * 256 + lane * 4, t);
      else *reinterpret_cast<f32x4*>(xn + tok * C + i4 * 256 + lane * 4) = t;
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// short-sequence multi-head attention (timm Attention == CoevoDecoder.py:118-131), N <= 32 tokens, 8 heads.
// One workgroup per sequence; K/V of the sequence staged once in LDS; wave w owns heads 2w, 2w+1; lane =
// (head-in-pair, query).  Online softmax over the <= 32 keys, all in registers; the K/V reads are
// half-wave broadcasts.  Sequence s, position i -> token (s % seq_div)*seq_lo + (s / seq_div)*seq_hi + i*tok_stride.
// ------------------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256) void seq_attention_kernel(const float* __restrict__ qkv, float* __restrict__ out, int N,
                                                            int seq_div, long long seq_lo, long long seq_hi,
                                                            long long tok_stride, int out_split) {
  constexpr int C = HD * 8;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ks = smem;
  float* Vs = smem + N * C;
  const int s = blockIdx.x;
  const long long base = (long long)(s % seq_div) * seq_lo + (long long)(s / seq_div) * seq_hi;
  const int tid = threadIdx.x;
  constexpr int C4 = C / 4;
  for (int idx = tid; idx < N * C4; idx += 256) {
    const int i = idx / C4, c = (idx % C4) * 4;
    const float* src = qkv + (base + i * tok_stride) * (3 * C);
    *reinterpret_cast<f32x4*>(Ks + i * C + c) = *reinterpret_cast<const f32x4*>(src + C + c);
    *reinterpret_cast<f32x4*>(Vs + i * C + c) = *reinterpret_cast<const f32x4*>(src + 2 * C + c);
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  const int head = wave * 2 + (lane >> 5), i = lane & 31;
  if (i >= N) return;
  const float scale = 1.44269504088896340736f / sqrtf((float)HD);  // hd^-0.5 * log2(e): softmax on the hardware 2^x
  float q[HD], o[HD];
  const float* qsrc = qkv + (base + i * tok_stride) * (3 * C) + head * HD;
#pragma unroll
  for (int d4 = 0; d4 < HD / 4; ++d4) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(qsrc + 4 * d4);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      q[4 * d4 + k] = t[k];
      o[4 * d4 + k] = 0.f;
    }
  }
  float m = -INFINITY, l = 0.f;
  for (int j = 0; j < N; ++j) {
    const float* kr = Ks + j * C + head * HD;
    float sc = 0.f;
#pragma unroll
    for (int d4 = 0; d4 < HD / 4; ++d4) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(kr + 4 * d4);
#pragma unroll
      for (int k = 0; k < 4; ++k) sc += q[4 * d4 + k] * t[k];
    }
    sc *= scale;
    const float mn = fmaxf(m, sc);
    const float corr = __builtin_amdgcn_exp2f(m - mn);  // first key: 2^(-inf) = 0
    const float pj = __builtin_amdgcn_exp2f(sc - mn);
    l = l * corr + pj;
    const float* vr = Vs + j * C + head * HD;
#pragma unroll
    for (int d4 = 0; d4 < HD / 4; ++d4) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(vr + 4 * d4);
#pragma unroll
      for (int k = 0; k < 4; ++k) o[4 * d4 + k] = o[4 * d4 + k] * corr + pj * t[k];
    }
    m = mn;
  }
  const float inv = 1.0f / l;
  float* dst = out + (base + i * tok_stride) * C;
#pragma unroll
  for (int d4 = 0; d4 < HD / 4; ++d4) {
    f32x4 t;
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = o[4 * d4 + k] * inv;
    if (out_split) store4_split_f16(dst, head * HD + 4 * d4, t);  // operand of a three-product f16 GEMM
    else *reinterpret_cast<f32x4*>(dst + head * HD + 4 * d4) = t;
  }
}

// ------------------------------------------------------------------------------------------------------
// The same attention for the sequence lengths the path actually runs (N = 16 frames, 17 or 19 joints), second version.
// The first kernel above is issue- and latency-bound, not memory-bound: one query per lane keeps 17 of every 32 lanes
// busy, the online softmax rescales the output at every key (3 FMA per key and channel), every lane re-reads all of K
// and V from LDS, and a sequence's K and V (70 KB at C = 512) leave room for two workgroups per CU only.
// Here a lane owns TWO queries of one head, so every K/V read from LDS serves two dot products and every multiply-add is
// a packed v_pk_fma_f32 over the query pair; N is a compile-time constant, so the N scores of a query live in registers
// and the softmax is two-pass (exact max, one exp per score, no rescaling: 2 FMA per key and channel).  All global
// traffic is full-row and coalesced: each of q, k, v of the sequence (C contiguous floats per token) is fetched into
// registers by all 128 threads and spread into LDS as [token][head][HD + 4]; the 4-float pad puts the 8 heads on disjoint
// bank groups, so a ds_read_b128 whose 16-lane service group spans several heads stays conflict-free.  ONE LDS region
// is used in turn for Q (until every lane holds its pair in registers), K (pass 1), V (pass 2) and the output tile:
// 37 KB per sequence at C = 512, four workgroups per CU, and the next array is already in registers (or in flight)
// when the region changes hands.  Used at C = 512 (HD = 64): 151 / 139 us per spatial / temporal launch against 195 /
// 179 us (B = 256); at C = 256 the first kernel already runs 16 waves per CU and both sit at ~4 TB/s, so it stays.
// ------------------------------------------------------------------------------------------------------
template <int HD, int N, bool OUT_SPLIT>
__global__ __launch_bounds__(128, 2) void seq_attention_pair_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                 int seq_div, long long seq_lo, long long seq_hi,
                                                                 long long tok_stride) {
  constexpr int C = HD * 8, HDP = HD + 4, C4 = C / 4, H4 = HD / 4;
  constexpr int NP = (N + 1) / 2;      // query pairs per head
  constexpr int CNT = N * C4;          // float4 per staged array (q, k or v of the whole sequence)
  constexpr int NIT = (CNT + 127) / 128;
  extern __shared__ __attribute__((aligned(16))) float R[];  // [N][8][HDP]: Q, then K, then V, then the output tile
  const int s = blockIdx.x, tid = threadIdx.x;
  const long long base = (long long)(s % seq_div) * seq_lo + (long long)(s / seq_div) * seq_hi;

  // element idx = tid + 128 u of a staged array is float4 c4 of token j; it lands at [j][head][HD + 4] in LDS
  auto lds_off = [&](int u) {
    const int idx = tid + 128 * u;
    const int j = idx / C4, c4 = idx % C4;
    return (j * 8 + c4 / H4) * HDP + 4 * (c4 % H4);
  };
  auto row_of = [&](int u) {  // tail threads re-read the last row (never stored)
    const int idx = tid + 128 * u;
    return qkv + (base + min(idx / C4, N - 1) * tok_stride) * (3 * C) + 4 * (idx % C4);
  };
  auto fetch_rows = [&](f32x4(&r)[NIT], int col) {
#pragma unroll
    for (int u = 0; u < NIT; ++u) r[u] = *reinterpret_cast<const f32x4*>(row_of(u) + col);
  };
  auto spread = [&](const f32x4(&r)[NIT]) {
#pragma unroll
    for (int u = 0; u < NIT; ++u)
      if (tid + 128 * u < CNT) *reinterpret_cast<f32x4*>(R + lds_off(u)) = r[u];
  };

  // ---- q, then k: every load of an array is issued before the first is used ----
  f32x4 ra[NIT], rb[NIT];
  fetch_rows(ra, 0);
  fetch_rows(rb, C);
  spread(ra);
  __syncthreads();

  // ---- this lane's query pair (dense over the 8 * NP pairs of the sequence) ----
  const bool active = tid < 8 * NP;
  const int h = active ? tid / NP : 0, pp = active ? tid % NP : 0;
  const int i0 = 2 * pp, i1 = min(2 * pp + 1, N - 1);
  // hd^-0.5 * log2(e): scores in log2 units, softmax on the hardware 2^x
  constexpr float scale = (HD == 32 ? 0.17677669529663688110f : 0.125f) * 1.44269504088896340736f;
  f32x2 qp[HD];
  if (active) {
#pragma unroll
    for (int d4 = 0; d4 < H4; ++d4) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(R + (i0 * 8 + h) * HDP + 4 * d4);
      const f32x4 b = *reinterpret_cast<const f32x4*>(R + (i1 * 8 + h) * HDP + 4 * d4);
#pragma unroll
      for (int k = 0; k < 4; ++k) qp[4 * d4 + k] = f32x2{a[k] * scale, b[k] * scale};
    }
  }
  __syncthreads();  // every lane holds its queries: the region takes K
  spread(rb);
  __syncthreads();

  // The two passes are software-pipelined by hand: the waves of a CU are few here (one or two per SIMD), so little but a
  // wave's own instruction order hides the ~100-cycle ds_read latency.  A step is 16 or 32 channels of one key: its four or
  // eight ds_read_b128 are issued one step ahead into the other half of a two-deep register buffer, and sched_barrier keeps
  // hipcc from sinking them back next to their uses (left alone it keeps two reads in flight and waits after every
  // fourth FMA).
  constexpr int RD = HD == 32 ? 8 : 4;  // ds_read_b128 per step (at HD = 64 the query pair leaves room for 2 x 4 only)
  constexpr int CPS = 4 * RD;           // channels per step
  constexpr int SPK = HD / CPS;         // steps per key
  constexpr int NST = N * SPK;
  f32x4 buf[2][RD];
  auto fetch = [&](int st, f32x4(&dst)[RD]) {
    const float* p = R + ((st / SPK) * 8 + h) * HDP + CPS * (st % SPK);
#pragma unroll
    for (int i = 0; i < RD; ++i) dst[i] = *reinterpret_cast<const f32x4*>(p + 4 * i);
  };

  // ---- pass 1: the N scores of both queries (four independent accumulator chains per key) ----
  f32x2 sc[N];
  if (active) {
    f32x2 a0, a1, a2, a3;
    fetch(0, buf[0]);
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      if (st + 1 < NST) fetch(st + 1, buf[(st + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      if (st % SPK == 0) a0 = a1 = a2 = a3 = f32x2{0.f, 0.f};
#pragma unroll
      for (int i = 0; i < RD; ++i) {
        const f32x4 k = buf[st & 1][i];
        const int d = CPS * (st % SPK) + 4 * i;
        a0 += qp[d + 0] * f32x2{k.x, k.x};
        a1 += qp[d + 1] * f32x2{k.y, k.y};
        a2 += qp[d + 2] * f32x2{k.z, k.z};
        a3 += qp[d + 3] * f32x2{k.w, k.w};
      }
      if (st % SPK == SPK - 1) sc[st / SPK] = (a0 + a1) + (a2 + a3);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  fetch_rows(ra, 2 * C);  // the queries are dead: v's loads fly under the softmax and the other workgroups' passes
  if (active) {
    // exact two-pass softmax over the N keys, per query
    f32x2 m = sc[0];
#pragma unroll
    for (int j = 1; j < N; ++j) m = f32x2{fmaxf(m.x, sc[j].x), fmaxf(m.y, sc[j].y)};
    f32x2 sum = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < N; ++j) {
      sc[j] = f32x2{__builtin_amdgcn_exp2f(sc[j].x - m.x), __builtin_amdgcn_exp2f(sc[j].y - m.y)};
      sum += sc[j];
    }
    const f32x2 inv = {1.0f / sum.x, 1.0f / sum.y};
#pragma unroll
    for (int j = 0; j < N; ++j) sc[j] *= inv;
  }
  __syncthreads();  // nobody reads K any more: the region takes V
  spread(ra);
  __syncthreads();

  // ---- pass 2: out = P V (HD independent accumulators per query) ----
  f32x2 o[HD];
  if (active) {
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = f32x2{0.f, 0.f};
    fetch(0, buf[0]);
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      if (st + 1 < NST) fetch(st + 1, buf[(st + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < RD; ++i) {
        const f32x4 v = buf[st & 1][i];
        const int d = CPS * (st % SPK) + 4 * i;
        o[d + 0] += sc[st / SPK] * f32x2{v.x, v.x};
        o[d + 1] += sc[st / SPK] * f32x2{v.y, v.y};
        o[d + 2] += sc[st / SPK] * f32x2{v.z, v.z};
        o[d + 3] += sc[st / SPK] * f32x2{v.w, v.w};
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __syncthreads();  // nobody reads V any more: the region takes the output tile, [token][head][HDP] like the inputs
  if (active) {
#pragma unroll
    for (int d4 = 0; d4 < H4; ++d4) {
      *reinterpret_cast<f32x4*>(R + (i0 * 8 + h) * HDP + 4 * d4) =
          f32x4{o[4 * d4].x, o[4 * d4 + 1].x, o[4 * d4 + 2].x, o[4 * d4 + 3].x};
      // odd N: the last pair's second query duplicates its first (i1 == i0), so this rewrites the same values - an
      // unconditional store (a predicated one makes hipcc spill hundreds of registers here)
      *reinterpret_cast<f32x4*>(R + (i1 * 8 + h) * HDP + 4 * d4) =
          f32x4{o[4 * d4].y, o[4 * d4 + 1].y, o[4 * d4 + 2].y, o[4 * d4 + 3].y};
    }
  }
  __syncthreads();
  // ---- LDS -> global, full rows ----
#pragma unroll
  for (int u = 0; u < NIT; ++u) {
    const int idx = tid + 128 * u;
    if (idx < CNT) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(R + lds_off(u));
      float* row = out + (base + (idx / C4) * tok_stride) * C;
      if constexpr (OUT_SPLIT) store4_split_f16(row, 4 * (idx % C4), t);
      else *reinterpret_cast<f32x4*>(row + 4 * (idx % C4)) = t;
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// regression head + frame fusion (PoseEstimation.py:62-66,109-113): one wavefront per (b, j):
//   p[t] = Wr * LN_{1e-5}(x[b,t,j,:]) + br ;  pose3d[b,j,:] = sum_t wf[t]*p[t] + bf
// ------------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void lifter_head_kernel(const float* __restrict__ x, const float* __restrict__ lnw,
                                                          const float* __restrict__ lnb, const float* __restrict__ Wr,
                                                          const float* __restrict__ br, const float* __restrict__ wf,
                                                          const float* __restrict__ bf, float* __restrict__ pose3d,
                                                          int B, int T, int J, const float* __restrict__ prew,
                                                          const float* __restrict__ preb, float pre_eps) {
  // One WORKGROUP per (b, j); wavefront w takes the frames t = w, w + 4, ... and has FOUR of their rows in flight before it touches the
  // first (round 6; until then one wavefront walked the T rows one after the other - sixteen dependent row fetches, 17 wavefronts per CU:
  // 54 us for a 142 MB read).  The per-frame results meet in LDS and are added in the order of t, as the serial loop added them.
  constexpr int NV = C / 64, RPW = 4;
  __shared__ float part[64][3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bj = blockIdx.x;
  const int b = bj / J, j = bj % J;
  // every per-channel parameter once per wavefront (fetched per row they are 14 KB through the vector cache for a 2 KB row)
  float pw[NV], pb[NV], hw[NV], hbv[NV], r0[NV], r1[NV], r2[NV];
  if (prew) {
    lane_channels<C>(prew, lane, pw);
    lane_channels<C>(preb, lane, pb);
  }
  lane_channels<C>(lnw, lane, hw);
  lane_channels<C>(lnb, lane, hbv);
  lane_channels<C>(Wr, lane, r0);
  lane_channels<C>(Wr + C, lane, r1);
  lane_channels<C>(Wr + 2 * C, lane, r2);
  for (int t0 = 0; t0 < T; t0 += 4 * RPW) {
    f32x4 raw[RPW][NV / 4];
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
      const int t = min(t0 + wave + 4 * k, T - 1);
      const float* row = x + ((long long)(b * T + t) * J + j) * C;
#pragma unroll
      for (int i4 = 0; i4 < NV / 4; ++i4) raw[k][i4] = *reinterpret_cast<const f32x4*>(row + i4 * 256 + lane * 4);
    }
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
      const int t = t0 + wave + 4 * k;
      float v[NV];
#pragma unroll
      for (int i4 = 0; i4 < NV / 4; ++i4)
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i4 * 4 + i] = raw[k][i4][i];
      if (prew) ln_regs_pre<C>(v, pw, pb, pre_eps);  // the last block's post-norm (norm_t), when no launch of its own ran it
      ln_regs_pre<C>(v, hw, hbv, 1e-5f);
      float d0 = 0.f, d1 = 0.f, d2 = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        d0 += v[i] * r0[i];
        d1 += v[i] * r1[i];
        d2 += v[i] * r2[i];
      }
      d0 = wave_sum(d0) + br[0];
      d1 = wave_sum(d1) + br[1];
      d2 = wave_sum(d2) + br[2];
      if (lane == 0 && t < T) {
        part[t][0] = d0;
        part[t][1] = d1;
        part[t][2] = d2;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    float acc = 0.f;
    for (int t = 0; t < T; ++t) acc += wf[t] * part[t][threadIdx.x];
    pose3d[(long long)bj * 3 + threadIdx.x] = acc + bf[0];
  }
}

// ------------------------------------------------------------------------------------------------------
// C-ABI launchers
// ------------------------------------------------------------------------------------------------------
extern "C" int pmce_embed_tokens_f32(const float* pose2d, const float* E, const float* Wje, const float* bje,
                                     const float* spos, float* x, long long ntok, int J, int C, hipStream_t stream) {
  PMCE_REQUIRE(C % 4 == 0 && J > 0 && ntok > 0, "embed_tokens: bad shape");
  const long long n = ntok * (C / 4);
  hipLaunchKernelGGL(embed_tokens_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, pose2d, E, Wje, bje, spos,
                     x, ntok, J, C);
  return pmce_check_launch("embed_tokens");
}

// The same followed by LayerNorm(w2, b2, eps2) of every token row in one launch: x = the tokens (fp32), xn = their LayerNorm (pre-split [row][C/16][16 hi |
// 16 lo*2^11] f16 when xn_split) - pmce_embed_tokens_f32 + pmce_ln_chain_ex_f32(out2) without the round trip of the tokens.  C = 256 or 512.
extern "C" int pmce_embed_ln_f32(const float* pose2d, const float* E, const float* Wje, const float* bje, const float* spos, float* x,
                                 long long ntok, int J, int C, const float* w2, const float* b2, float eps2, float* xn, int xn_split,
                                 hipStream_t stream) {
  PMCE_REQUIRE(pose2d && E && Wje && bje && spos && x && w2 && b2 && xn, "embed_ln: null pointer");
  PMCE_REQUIRE((C == 256 || C == 512) && J > 0 && ntok > 0 && ntok % J == 0 && ntok < (1ll << 31),
               "embed_ln: C must be 256 or 512, J positive, ntok a positive multiple of J (whole frames) below 2^31");
  const long long nframes = ntok / J;
  // wavefronts per frame: enough wavefronts for the chip (8 per SIMD of 256 CUs), as few as that allows
  long long wpf = (8192 + nframes - 1) / nframes;
  wpf = wpf < 1 ? 1 : (wpf > J ? J : wpf);
  const unsigned grid = (unsigned)((nframes * wpf + 3) / 4);
  if (C == 256)
    hipLaunchKernelGGL((embed_ln_kernel<256>), dim3(grid), dim3(256), 0, stream, pose2d, E, Wje, bje, spos, x, nframes, J, w2, b2, eps2, xn, xn_split, (int)wpf);
  else
    hipLaunchKernelGGL((embed_ln_kernel<512>), dim3(grid), dim3(256), 0, stream, pose2d, E, Wje, bje, spos, x, nframes, J, w2, b2, eps2, xn, xn_split, (int)wpf);
  return pmce_check_launch("embed_ln");
}

extern "C" int pmce_ln_chain_ex_f32(const float* x, long long rows, int C, const float* w1, const float* b1, float eps1,
                                    const float* add, int add_div, int add_mod, float* out1, const float* w2, const float* b2,
                                    float eps2, float* out2, int out2_split, hipStream_t stream) {
  PMCE_REQUIRE(C == 256 || C == 512, "ln_chain: C must be 256 or 512 (got %d)", C);
  PMCE_REQUIRE(rows > 0 && (out1 || out2), "ln_chain: nothing to do");
  PMCE_REQUIRE(!out2 || (w2 && b2), "ln_chain: out2 needs w2/b2");
  if (add_div <= 0) add_div = 1;
  if (add_mod <= 0) add_mod = 1;
  const unsigned grid = (unsigned)((rows + 3) / 4);
  if (C == 256)
    hipLaunchKernelGGL((ln_chain_kernel<256>), dim3(grid), dim3(256), 0, stream, x, rows, w1, b1, eps1, add, add_div, add_mod,
                       out1, w2, b2, eps2, out2, nullptr, 0, out2_split);
  else
    hipLaunchKernelGGL((ln_chain_kernel<512>), dim3(grid), dim3(256), 0, stream, x, rows, w1, b1, eps1, add, add_div, add_mod,
                       out1, w2, b2, eps2, out2, nullptr, 0, out2_split);
  return pmce_check_launch("ln_chain");
}
extern "C" int pmce_ln_chain_f32(const float* x, long long rows, int C, const float* w1, const float* b1, float eps1,
                                 const float* add, int add_div, int add_mod, float* out1, const float* w2, const float* b2,
                                 float eps2, float* out2, hipStream_t stream) {
  return pmce_ln_chain_ex_f32(x, rows, C, w1, b1, eps1, add, add_div, add_mod, out1, w2, b2, eps2, out2, 0, stream);
}

// Streaming: tokens of W windows from the per-frame table x0[L,J,C] (= norm_s(SpatialBlocks[0](embed)), window-independent):
//   X[w,t,j,:] = x0[frame(w,t),j,:] + tpos[t,:]  (PoseEstimation.py:87-88) ;  XN = LN(X; w2,b2,eps2)  (TemporalBlocks[0].norm1)
extern "C" int pmce_window_tokens_ex_f32(const float* x0, const int* win, const float* tpos, const float* w2, const float* b2,
                                         float eps2, float* X, float* XN, int W, int L, int T, int J, int C, int xn_split,
                                         hipStream_t stream) {
  PMCE_REQUIRE(C == 256 || C == 512, "window_tokens: C must be 256 or 512");
  PMCE_REQUIRE(x0 && win && tpos && w2 && b2 && X && XN && W > 0 && L > 0 && T > 0 && J > 0, "window_tokens: bad args");
  const long long rows = (long long)W * T * J;
  const unsigned grid = (unsigned)((rows + 3) / 4);
  if (C == 256)
    hipLaunchKernelGGL((ln_chain_kernel<256>), dim3(grid), dim3(256), 0, stream, x0, rows, nullptr, nullptr, 0.f, tpos, J, T, X,
                       w2, b2, eps2, XN, win, L, xn_split);
  else
    hipLaunchKernelGGL((ln_chain_kernel<512>), dim3(grid), dim3(256), 0, stream, x0, rows, nullptr, nullptr, 0.f, tpos, J, T, X,
                       w2, b2, eps2, XN, win, L, xn_split);
  return pmce_check_launch("window_tokens");
}
extern "C" int pmce_window_tokens_f32(const float* x0, const int* win, const float* tpos, const float* w2, const float* b2,
                                      float eps2, float* X, float* XN, int W, int L, int T, int J, int C,
                                      hipStream_t stream) {
  return pmce_window_tokens_ex_f32(x0, win, tpos, w2, b2, eps2, X, XN, W, L, T, J, C, 0, stream);
}

// Streaming: time-major gather of per-frame rows:  dst[t][w][:] = src[frame(w,t)][:]   (ncols % 4 == 0)
__global__ __launch_bounds__(256) void window_rows_kernel(const float* __restrict__ src, const int* __restrict__ win,
                                                          float* __restrict__ dst, int W, int L, int T, int ncols) {
  const int tw = blockIdx.x;  // t * W + w
  const int t = tw / W, w = tw % W;
  const int s0 = win[2 * w], e0 = win[2 * w + 1];
  const int fr = min(max(s0 == e0 ? s0 : s0 + t, 0), L - 1);
  const f32x4* s = reinterpret_cast<const f32x4*>(src + (long long)fr * ncols);
  f32x4* d = reinterpret_cast<f32x4*>(dst + (long long)tw * ncols);
  for (int i = threadIdx.x; i < ncols / 4; i += 256) d[i] = s[i];
}

extern "C" int pmce_window_rows_f32(const float* src, const int* win, float* dst, int W, int L, int T, int ncols,
                                    hipStream_t stream) {
  PMCE_REQUIRE(src && win && dst && W > 0 && L > 0 && T > 0 && ncols > 0 && ncols % 4 == 0, "window_rows: bad args");
  hipLaunchKernelGGL(window_rows_kernel, dim3(T * W), dim3(256), 0, stream, src, win, dst, W, L, T, ncols);
  return pmce_check_launch("window_rows");
}

template <int HD, int N>
static int launch_seq_attention_pair(const float* qkv, float* out, int nseq, int seq_div, long long seq_lo, long long seq_hi,
                                     long long tok_stride, int out_split, hipStream_t stream) {
  constexpr int lds = N * 8 * (HD + 4) * (int)sizeof(float);
  if (lds > 65536) {
    static std::atomic<unsigned long long> attr{0}, attr_s{0};
    PMCE_TRY(pmce_opt_in_lds((const void*)seq_attention_pair_kernel<HD, N, false>, lds, attr, "seq_attention"));
    PMCE_TRY(pmce_opt_in_lds((const void*)seq_attention_pair_kernel<HD, N, true>, lds, attr_s, "seq_attention"));
  }
  if (out_split)
    hipLaunchKernelGGL((seq_attention_pair_kernel<HD, N, true>), dim3(nseq), dim3(128), lds, stream, qkv, out, seq_div, seq_lo,
                       seq_hi, tok_stride);
  else
    hipLaunchKernelGGL((seq_attention_pair_kernel<HD, N, false>), dim3(nseq), dim3(128), lds, stream, qkv, out, seq_div, seq_lo,
                       seq_hi, tok_stride);
  return pmce_check_launch("seq_attention");
}

extern "C" int pmce_seq_attention_ex_f32(const float* qkv, float* out, int nseq, int N, int C, int seq_div, long long seq_lo,
                                         long long seq_hi, long long tok_stride, int out_split, hipStream_t stream) {
  PMCE_REQUIRE(C == 256 || C == 512, "seq_attention: C must be 256 or 512 (8 heads of 32/64)");
  PMCE_REQUIRE(N >= 1 && N <= 32 && nseq > 0, "seq_attention: N must be in 1..32 (got %d)", N);
  if (seq_div <= 0) seq_div = 0x7fffffff;
  {  // the sequence lengths of the path: 16 frames, 17 (H36M) or 19 (COCO + pelvis, neck) joints
    // C = 512 only: at C = 256 both kernels sit at the same ~4 TB/s (the first one already has 16 waves per CU there)
#define PMCE_PAIR(HD_, N_) \
  if (C == 8 * HD_ && N == N_) return launch_seq_attention_pair<HD_, N_>(qkv, out, nseq, seq_div, seq_lo, seq_hi, tok_stride, out_split, stream)
    PMCE_PAIR(64, 16);
    PMCE_PAIR(64, 17);
    PMCE_PAIR(64, 19);
#undef PMCE_PAIR
  }
  const size_t lds = (size_t)2 * N * C * sizeof(float);
  if (C == 256) {
    static std::atomic<unsigned long long> attr256{0};
    PMCE_TRY(pmce_opt_in_lds((const void*)seq_attention_kernel<32>, 65536, attr256, "seq_attention"));
    hipLaunchKernelGGL((seq_attention_kernel<32>), dim3(nseq), dim3(256), lds, stream, qkv, out, N, seq_div, seq_lo, seq_hi,
                       tok_stride, out_split);
  } else {
    static std::atomic<unsigned long long> attr512{0};
    PMCE_TRY(pmce_opt_in_lds((const void*)seq_attention_kernel<64>, 131072, attr512, "seq_attention"));
    hipLaunchKernelGGL((seq_attention_kernel<64>), dim3(nseq), dim3(256), lds, stream, qkv, out, N, seq_div, seq_lo, seq_hi,
                       tok_stride, out_split);
  }
  return pmce_check_launch("seq_attention");
}
extern "C" int pmce_seq_attention_f32(const float* qkv, float* out, int nseq, int N, int C, int seq_div, long long seq_lo,
                                      long long seq_hi, long long tok_stride, hipStream_t stream) {
  return pmce_seq_attention_ex_f32(qkv, out, nseq, N, C, seq_div, seq_lo, seq_hi, tok_stride, 0, stream);
}

// prew != null: x holds the LAST TemporalBlock's output BEFORE its post-norm; the rows pass through LayerNorm(prew, preb, pre_eps) (norm_t,
// PoseEstimation.py:92) on the way in - the head then needs no ln_chain launch (and no 2 x 143 MB round trip at B = 256, C = 512) in front of it.
extern "C" int pmce_lifter_head_ex_f32(const float* x, const float* prew, const float* preb, float pre_eps, const float* lnw,
                                       const float* lnb, const float* Wr, const float* br, const float* wf, const float* bf,
                                       float* pose3d, int B, int T, int J, int C, hipStream_t stream) {
  PMCE_REQUIRE(C == 256 || C == 512, "lifter_head: C must be 256 or 512");
  PMCE_REQUIRE((prew == nullptr) == (preb == nullptr), "lifter_head: the pre-norm needs weight and bias");
  PMCE_REQUIRE(T > 0 && T <= 64, "lifter_head: 1..64 frames per clip");
  const unsigned grid = (unsigned)(B * J);
  const unsigned threads = 256u;
  if (C == 256)
    hipLaunchKernelGGL((lifter_head_kernel<256>), dim3(grid), dim3(threads), 0, stream, x, lnw, lnb, Wr, br, wf, bf, pose3d, B, T, J, prew, preb, pre_eps);
  else
    hipLaunchKernelGGL((lifter_head_kernel<512>), dim3(grid), dim3(threads), 0, stream, x, lnw, lnb, Wr, br, wf, bf, pose3d, B, T, J, prew, preb, pre_eps);
  return pmce_check_launch("lifter_head");
}
extern "C" int pmce_lifter_head_f32(const float* x, const float* lnw, const float* lnb, const float* Wr, const float* br,
                                    const float* wf, const float* bf, float* pose3d, int B, int T, int J, int C,
                                    hipStream_t stream) {
  return pmce_lifter_head_ex_f32(x, nullptr, nullptr, 0.f, lnw, lnb, Wr, br, wf, bf, pose3d, B, T, J, C, stream);
}
