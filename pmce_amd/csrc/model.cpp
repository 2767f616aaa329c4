// Host-side orchestration of the PMCE hot path: a descriptor of registered device pointers plus the launch
// sequences of GraphormerNet.forward (reference lib/models/PoseEstimation.py:95-115), Pose2Mesh.forward
// (lib/models/CoevoDecoder.py:226-246), PMCE.forward (lib/models/PMCE.py:15-20) and the caller's J_regressor
// projection (lib/core/base.py:223-225).  No device memory is allocated here: weights and workspace belong to
// the caller.  Everything is launched on the caller's stream, in order, with no host synchronisation, so a
// forward is hipGraph-capturable.
#include <hip/hip_runtime.h>

#include <stdlib.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "common.hpp"
#include "../../include/pmce_hip.h"

namespace {

constexpr int T = 16, F = 2048, NVC = 431, NVF = 6890, D = 64, GH = 1024;
constexpr int FINAL_K = 3360;  // 2048 + 1293 = 3341 rounded up to a multiple of 32
constexpr int N_ADA = 24;      // live AdaLN instances (joint stream of blocks 1-2 is dead at inference)

enum ProfClass {
  P_GEMM_LIFTER, P_GEMM_GRU_IN, P_GRU_STEP, P_GEMM_ADA, P_GEMM_FINAL, P_LN, P_SEQ_ATTN, P_EMBED, P_HEAD,
  P_GATHER, P_JOINT_EMBED, P_CA_FOLD, P_VERTEX_CA, P_ADALN_MLP, P_ADALN_QKV, P_VERTEX_SA, P_TOKENS_KV, P_JOINT_STREAM,
  P_FINAL_OP, P_JREG, P_MISC, P_COUNT
};
const char* kProfNames[P_COUNT] = {
    "gemm_lifter", "gemm_gru_in", "gru_step", "gemm_ada", "gemm_final", "ln_chain", "seq_attention", "embed_tokens",
    "lifter_head", "vertex_init_gather", "joint_embed", "ca_fold", "vertex_ca", "adaln_mlp", "adaln_qkv",
    "vertex_sa", "tokens_kv", "joint_stream", "build_final_operand", "j_regress", "misc"};

struct Ev {
  hipEvent_t a, b;
  int cls;
};

}  // namespace

struct pmce_model {
  int J, C, depth;
  std::vector<std::string> names;
  std::unordered_map<std::string, const void*> ptr;
  bool finalized = false;
  bool has_lifter = false, has_decoder = false;
  // second stream for the image-feature branch (created on first use, destroyed with the model)
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_a = nullptr, ev_b = nullptr, ev_c = nullptr, ev_d = nullptr;
  hipEvent_t ev_lifter = nullptr;  // recorded by pmce_forward when its pose lifter is enqueued (pmce_model_wait_lifter)
  bool concurrent = true;  // pmce_model_set_concurrency
  // regressor (optional)
  const int* jr_indptr = nullptr;
  const int* jr_indices = nullptr;
  const float* jr_data = nullptr;
  int jr_rows = 0;
  // profiling
  bool prof = false;
  std::vector<Ev> pending;
  std::vector<Ev> pool;
  double prof_ms[P_COUNT] = {0};
  long long prof_n[P_COUNT] = {0};

  const float* f(const std::string& n) const { return static_cast<const float*>(ptr.at(n)); }
  const int* i32(const std::string& n) const { return static_cast<const int*>(ptr.at(n)); }
};

namespace {

struct ProfScope {
  pmce_model* m;
  hipStream_t s;
  Ev e;
  bool on;
  ProfScope(pmce_model* m_, int cls, hipStream_t s_) : m(m_), s(s_), on(m_->prof) {
    if (!on) return;
    if (!m->pool.empty()) {
      e = m->pool.back();
      m->pool.pop_back();
    } else {
      (void)hipEventCreate(&e.a);
      (void)hipEventCreate(&e.b);
    }
    e.cls = cls;
    (void)hipEventRecord(e.a, s);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(e.b, s);
    m->pending.push_back(e);
  }
};

#define RUN(cls, call)             \
  do {                             \
    ProfScope _ps(m, cls, stream); \
    PMCE_TRY(call);                \
  } while (0)

std::string blk(const char* kind, int i, const char* leaf) {
  return std::string("lifter.") + kind + "Blocks." + std::to_string(i) + "." + leaf;
}

void build_names(pmce_model* m) {
  auto& n = m->names;
  for (const char* s : {"lifter.joint_embed.weight", "lifter.joint_embed.bias", "lifter.imgfeat_embed.weight",
                        "lifter.imgfeat_embed.bias", "lifter.spatial_pos_embed", "lifter.temporal_pos_embed"})
    n.push_back(s);
  for (const char* kind : {"Spatial", "Temporal"})
    for (int i = 0; i < m->depth; ++i)
      for (const char* leaf : {"norm1.weight", "norm1.bias", "attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight",
                               "attn.proj.bias", "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias",
                               "mlp.fc2.weight", "mlp.fc2.bias"})
        n.push_back(blk(kind, i, leaf));
  for (const char* s : {"lifter.norm_s.weight", "lifter.norm_s.bias", "lifter.norm_t.weight", "lifter.norm_t.bias",
                        "lifter.regression.0.weight", "lifter.regression.0.bias", "lifter.regression.1.weight",
                        "lifter.regression.1.bias", "lifter.fusion.weight", "lifter.fusion.bias"})
    n.push_back(s);
  for (const char* s : {"dec.vj_relation", "dec.gru.w_ih_l0", "dec.gru.b_ih_l0", "dec.gru.w_hh_l0", "dec.gru.b_hh_l0",
                        "dec.gru.w_ih_l1", "dec.gru.b_ih_l1", "dec.gru.w_hh_l1", "dec.gru.b_hh_l1", "dec.ada.weight",
                        "dec.ada.bias"})
    n.push_back(s);
  for (int k = 1; k <= 3; ++k) {
    const std::string p = "dec.b" + std::to_string(k) + ".";
    for (const char* leaf :
         {"joint_proj.weight", "joint_proj.bias", "joint_pos_embed", "proj_j2v_dim.weight", "proj_j2v_dim.bias",
          "j2v_K_embed", "vertx_proj.weight", "Eq", "vca.wq.weight", "vca.wq.bias", "vca.wk.weight", "vca.wk.bias",
          "vca.wv.weight", "vca.wv.bias", "vca.proj.weight", "vca.proj.bias", "vca.mlp.fc1.weight", "vca.mlp.fc1.bias",
          "vca.mlp.fc2.weight", "vca.mlp.fc2.bias", "vsa.qkv.weight", "vsa.qkv.bias", "vsa.proj.weight", "vsa.proj.bias",
          "vsa.mlp.fc1.weight", "vsa.mlp.fc1.bias", "vsa.mlp.fc2.weight", "vsa.mlp.fc2.bias", "vcoor.weight", "vcoor.bias"})
      n.push_back(p + leaf);
  }
  for (const char* leaf :
       {"Ev", "proj_v2j_dim.weight", "Ek", "j_Q_embed", "jca.wq.weight", "jca.wq.bias", "jca.wk.weight", "jca.wk.bias",
        "jca.wv.weight", "jca.wv.bias", "jca.proj.weight", "jca.proj.bias", "jca.mlp.fc1.weight", "jca.mlp.fc1.bias",
        "jca.mlp.fc2.weight", "jca.mlp.fc2.bias", "jsa.qkv.weight", "jsa.qkv.bias", "jsa.proj.weight", "jsa.proj.bias",
        "jsa.mlp.fc1.weight", "jsa.mlp.fc1.bias", "jsa.mlp.fc2.weight", "jsa.mlp.fc2.bias", "jcoor.weight", "jcoor.bias"})
    n.push_back(std::string("dec.b3.") + leaf);
  n.push_back("dec.final.weight");
  n.push_back("dec.final.bias");
}

// ---- workspace carving ---------------------------------------------------------------------------------
struct Carver {
  char* base;
  size_t off = 0, cap;
  Carver(void* b, size_t c) : base(static_cast<char*>(b)), cap(c) {}
  float* take(size_t nfloat) {
    float* p = reinterpret_cast<float*>(base + off);
    off += ((nfloat * sizeof(float) + 255) / 256) * 256;
    return p;
  }
};

struct LifterWs {
  float *E, *X, *XN, *QKV, *AO;
};
struct DecoderWs {
  float *GI0, *Y0, *GI1, *Y1, *GB, *VT[3], *JF[3], *XK[3], *KF[3], *S0[3], *VF[3], *F1, *F2, *QKV, *KVJ, *FA, *JM;
};

void carve_lifter(Carver& c, const pmce_model* m, int B, LifterWs& w) {
  const size_t M = (size_t)B * T * m->J, C = m->C;
  w.E = c.take((size_t)B * T * C);
  w.X = c.take(M * C);
  w.XN = c.take(M * C);
  w.QKV = c.take(M * 3 * C);  // also the MLP hidden [M,2C]
  w.AO = c.take(M * C);
}
void carve_decoder(Carver& c, const pmce_model* m, int B, DecoderWs& w) {
  w.GI0 = c.take((size_t)T * B * 6 * GH);
  w.Y0 = c.take((size_t)T * B * 2 * GH);
  w.GI1 = c.take((size_t)2 * 9 * B * 3 * GH);
  w.Y1 = c.take((size_t)T * B * 2 * GH);
  w.GB = c.take((size_t)B * N_ADA * 128);
  for (int i = 0; i < 3; ++i) w.VT[i] = c.take((size_t)B * NVC * 3);
  for (int i = 0; i < 3; ++i) {
    w.JF[i] = c.take((size_t)B * 32 * D);
    w.XK[i] = c.take((size_t)B * 32 * D);
    w.KF[i] = c.take((size_t)B * 4096);
    w.S0[i] = c.take((size_t)B * 64);
    w.VF[i] = c.take((size_t)B * 4096);
  }
  w.F1 = c.take((size_t)B * NVC * D);
  w.F2 = c.take((size_t)B * NVC * D);
  w.QKV = c.take((size_t)B * NVC * 3 * D);
  w.KVJ = c.take((size_t)B * NVC * 2 * D);
  w.FA = c.take((size_t)B * FINAL_K);
  w.JM = c.take((size_t)B * 32 * 3);
}

int gemm(const float* A, const float* W, const float* bias, const float* R, float* Cc, int M, int N, int K, long long lda,
         long long ldc, int act, hipStream_t s) {
  return pmce_gemm_nt_f32(A, W, bias, R, Cc, M, N, K, lda, K, ldc, act, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, s);
}

// ---- GraphormerNet.forward --------------------------------------------------------------------------------
// Everything up to and including SpatialBlocks[0] is PER FRAME (its attention runs over the J joints of one frame,
// PoseEstimation.py:78-85), so it is written over `nframes` frames: B*16 for independent clips, L for a streamed sequence.

// attention + MLP of one block (pre-norm input in w.XN, residual stream in w.X); kind 0 spatial, 1 temporal
int lifter_block_body(pmce_model* m, int kind, int i, long long M, int nframes, int B, LifterWs& w, hipStream_t stream) {
  const int J = m->J, C = m->C;
  const char* kn = kind == 0 ? "Spatial" : "Temporal";
  RUN(P_GEMM_LIFTER, gemm(w.XN, m->f(blk(kn, i, "attn.qkv.weight")), m->f(blk(kn, i, "attn.qkv.bias")), nullptr, w.QKV,
                          (int)M, 3 * C, C, C, 3 * C, 0, stream));
  if (kind == 0)  // sequences = frames, tokens j contiguous                        (PoseEstimation.py:78,101)
    RUN(P_SEQ_ATTN, pmce_seq_attention_f32(w.QKV, w.AO, nframes, J, C, 0, J, 0, 1, stream));
  else  // sequences = (b,j), tokens t at stride J                                  (PoseEstimation.py:87,104)
    RUN(P_SEQ_ATTN, pmce_seq_attention_f32(w.QKV, w.AO, B * J, T, C, J, 1, (long long)T * J, J, stream));
  RUN(P_GEMM_LIFTER, gemm(w.AO, m->f(blk(kn, i, "attn.proj.weight")), m->f(blk(kn, i, "attn.proj.bias")), w.X, w.X, (int)M, C,
                          C, C, C, 0, stream));
  RUN(P_LN, pmce_ln_chain_f32(w.X, M, C, nullptr, nullptr, 0.f, nullptr, 1, 1, nullptr, m->f(blk(kn, i, "norm2.weight")),
                              m->f(blk(kn, i, "norm2.bias")), 1e-6f, w.XN, stream));
  float* Hid = w.QKV;
  RUN(P_GEMM_LIFTER, gemm(w.XN, m->f(blk(kn, i, "mlp.fc1.weight")), m->f(blk(kn, i, "mlp.fc1.bias")), nullptr, Hid, (int)M,
                          2 * C, C, C, 2 * C, 1, stream));
  RUN(P_GEMM_LIFTER, gemm(Hid, m->f(blk(kn, i, "mlp.fc2.weight")), m->f(blk(kn, i, "mlp.fc2.bias")), w.X, w.X, (int)M, C,
                          2 * C, 2 * C, C, 0, stream));
  return PMCE_OK;
}

// embedding + SpatialBlocks[0] over `nframes` frames; leaves the block output (before norm_s) in w.X
int lifter_frames(pmce_model* m, const float* pose2d, const float* img_feat, int nframes, LifterWs& w, hipStream_t stream) {
  const int J = m->J, C = m->C;
  const long long M = (long long)nframes * J;
  PMCE_REQUIRE(M < (1ll << 31), "lifter: too many tokens");
  RUN(P_GEMM_LIFTER, gemm(img_feat, m->f("lifter.imgfeat_embed.weight"), m->f("lifter.imgfeat_embed.bias"), nullptr, w.E,
                          nframes, C, F, F, C, 0, stream));
  RUN(P_EMBED, pmce_embed_tokens_f32(pose2d, w.E, m->f("lifter.joint_embed.weight"), m->f("lifter.joint_embed.bias"),
                                     m->f("lifter.spatial_pos_embed"), w.X, M, J, C, stream));
  RUN(P_LN, pmce_ln_chain_f32(w.X, M, C, nullptr, nullptr, 0.f, nullptr, 1, 1, nullptr,
                              m->f(blk("Spatial", 0, "norm1.weight")), m->f(blk("Spatial", 0, "norm1.bias")), 1e-6f, w.XN,
                              stream));
  return lifter_block_body(m, 0, 0, M, nframes, 0, w, stream);
}

// post-norm of block (kind, i): norm_s / norm_t (shared across depth, eps 1e-6), + temporal_pos_embed after the first
// spatial block, fused with the NEXT block's norm1
int lifter_post_norm(pmce_model* m, int kind, int i, long long M, LifterWs& w, hipStream_t stream) {
  const int J = m->J, C = m->C;
  const float* nw = m->f(kind == 0 ? "lifter.norm_s.weight" : "lifter.norm_t.weight");
  const float* nb = m->f(kind == 0 ? "lifter.norm_s.bias" : "lifter.norm_t.bias");
  const float* add = (kind == 0 && i == 0) ? m->f("lifter.temporal_pos_embed") : nullptr;
  const float *w2 = nullptr, *b2 = nullptr;
  if (kind == 0) {
    w2 = m->f(blk("Temporal", i, "norm1.weight"));
    b2 = m->f(blk("Temporal", i, "norm1.bias"));
  } else if (i + 1 < m->depth) {
    w2 = m->f(blk("Spatial", i + 1, "norm1.weight"));
    b2 = m->f(blk("Spatial", i + 1, "norm1.bias"));
  }
  RUN(P_LN, pmce_ln_chain_f32(w.X, M, C, nw, nb, 1e-6f, add, J, T, w.X, w2, b2, 1e-6f, w2 ? w.XN : nullptr, stream));
  return PMCE_OK;
}

// everything from TemporalBlocks[0] on, for B clips whose tokens (after norm_s + temporal_pos_embed) are in w.X and whose
// TemporalBlocks[0].norm1 output is in w.XN
int lifter_rest(pmce_model* m, float* pose3d, int B, LifterWs& w, hipStream_t stream) {
  const int J = m->J, C = m->C;
  const long long M = (long long)B * T * J;
  for (int i = 0; i < m->depth; ++i) {
    for (int kind = (i == 0 ? 1 : 0); kind < 2; ++kind) {
      PMCE_TRY(lifter_block_body(m, kind, i, M, B * T, B, w, stream));
      PMCE_TRY(lifter_post_norm(m, kind, i, M, w, stream));
    }
  }
  RUN(P_HEAD, pmce_lifter_head_f32(w.X, m->f("lifter.regression.0.weight"), m->f("lifter.regression.0.bias"),
                                   m->f("lifter.regression.1.weight"), m->f("lifter.regression.1.bias"),
                                   m->f("lifter.fusion.weight"), m->f("lifter.fusion.bias"), pose3d, B, T, J, C, stream));
  return PMCE_OK;
}

int lifter_impl(pmce_model* m, const float* pose2d, const float* img_feat, float* pose3d, int B, LifterWs& w,
                hipStream_t stream) {
  PMCE_TRY(lifter_frames(m, pose2d, img_feat, B * T, w, stream));
  PMCE_TRY(lifter_post_norm(m, 0, 0, (long long)B * T * m->J, w, stream));
  return lifter_rest(m, pose3d, B, w, stream);
}

// ---- Pose2Mesh.forward ------------------------------------------------------------------------------------
int gru_layer(pmce_model* m, int layer, const float* gi_f, const float* gi_b, long long gi_rs, int t_f0, int t_b0,
              int nsteps_f, int nsteps_b, float* Y, int B, hipStream_t stream) {
  // direction 0 walks t = t_f0, t_f0+1, ...; direction 1 walks t = t_b0, t_b0-1, ...  Y is [T][B][2*GH].
  const std::string l = std::to_string(layer);
  const float* whh = m->f("dec.gru.w_hh_l" + l);
  const float* bhh = m->f("dec.gru.b_hh_l" + l);
  const long long YS = (long long)B * 2 * GH;
  const int nsteps = nsteps_f > nsteps_b ? nsteps_f : nsteps_b;
  for (int s = 0; s < nsteps; ++s) {
    const bool af = s < nsteps_f, ab = s < nsteps_b;
    const int tf = t_f0 + s, tb = t_b0 - s;
    const float* hp_f = (s > 0 && af) ? Y + (long long)(tf - 1) * YS : nullptr;
    const float* hp_b = (s > 0 && ab) ? Y + (long long)(tb + 1) * YS + GH : nullptr;
    const float* gif = gi_f + (long long)s * B * gi_rs;                   // slab of time tf
    const float* gib = gi_b + (long long)(nsteps_b - 1 - s) * B * gi_rs;  // slab of time tb (slabs stored ascending in t)
    float* ho_f = Y + (long long)tf * YS;
    float* ho_b = Y + (long long)tb * YS + GH;
    const float* whh_b = whh + (long long)3 * GH * GH;
    const float* bhh_b = bhh + 3 * GH;
    if (af && ab)
      RUN(P_GRU_STEP, pmce_gru_step_f32(gif, gib, whh, whh_b, bhh, bhh_b, hp_f, hp_b, ho_f, ho_b, gi_rs, 2 * GH, B, GH, 2, stream));
    else if (af)
      RUN(P_GRU_STEP, pmce_gru_step_f32(gif, nullptr, whh, nullptr, bhh, nullptr, hp_f, nullptr, ho_f, nullptr, gi_rs, 2 * GH, B, GH,
                                        1, stream));
    else
      RUN(P_GRU_STEP, pmce_gru_step_f32(gib, nullptr, whh_b, nullptr, bhh_b, nullptr, hp_b, nullptr, ho_b, nullptr, gi_rs, 2 * GH, B,
                                        GH, 1, stream));
  }
  return PMCE_OK;
}

// image-feature branch of Pose2Mesh.forward: bi-GRU + AdaLN parameters.  Depends only on img_feat, so pmce_forward
// runs it on a second stream concurrently with the pose lifter.
// recurrences + AdaLN parameters, given the layer-0 input projections GI0 (time-major [t][b][6144])
int gru_rest(pmce_model* m, int B, DecoderWs& w, hipStream_t stream);

int gru_part(pmce_model* m, const float* img_feat, int B, DecoderWs& w, hipStream_t stream) {
  // ---- bi-GRU over the 16 frames (CoevoDecoder.py:228); buffers are time-major [t][b][.] ----
  // layer 0 input projections for both directions in one product: rows (b,t) of img_feat -> rows (t,b) of GI0
  RUN(P_GEMM_GRU_IN, pmce_gemm_nt_f32(img_feat, m->f("dec.gru.w_ih_l0"), m->f("dec.gru.b_ih_l0"), nullptr, w.GI0, B * T,
                                      6 * GH, F, F, F, 6 * GH, 0, 0, 0, 0, T, (long long)B * 6 * GH, 6 * GH, 1, 0, 0, 0, 0,
                                      stream));
  return gru_rest(m, B, w, stream);
}

int gru_rest(pmce_model* m, int B, DecoderWs& w, hipStream_t stream) {
  // layer 0: both directions over all 16 steps.  gi slabs: fwd reads column block 0, bwd column block 1 of GI0.
  PMCE_TRY(gru_layer(m, 0, w.GI0, w.GI0 + 3 * GH, 6 * GH, 0, T - 1, T, T, w.Y0, B, stream));
  // layer 1: only y[8] is consumed (CoevoDecoder.py:229,241-243) -> fwd needs t = 0..8, bwd t = 15..8.
  float* GI1f = w.GI1;
  float* GI1b = w.GI1 + (long long)9 * B * 3 * GH;
  RUN(P_GEMM_GRU_IN, gemm(w.Y0, m->f("dec.gru.w_ih_l1"), m->f("dec.gru.b_ih_l1"), nullptr, GI1f, 9 * B, 3 * GH, 2 * GH, 2 * GH,
                          3 * GH, 0, stream));
  RUN(P_GEMM_GRU_IN, gemm(w.Y0 + (long long)8 * B * 2 * GH, m->f("dec.gru.w_ih_l1") + (long long)3 * GH * 2 * GH,
                          m->f("dec.gru.b_ih_l1") + 3 * GH, nullptr, GI1b, 8 * B, 3 * GH, 2 * GH, 2 * GH, 3 * GH, 0, stream));
  PMCE_TRY(gru_layer(m, 1, GI1f, GI1b, 3 * GH, 0, T - 1, 9, 8, w.Y1, B, stream));
  const float* g = w.Y1 + (long long)8 * B * 2 * GH;  // img_feat = y[seqlen // 2], [B, 2048]

  // ---- all live AdaLN gamma/beta in one product (CoevoDecoder.py:19-20,27-28) ----
  RUN(P_GEMM_ADA, gemm(g, m->f("dec.ada.weight"), m->f("dec.ada.bias"), nullptr, w.GB, B, N_ADA * 128, 2 * GH, 2 * GH,
                       N_ADA * 128, 0, stream));
  return PMCE_OK;
}

// joint-side preparation of CoevoBlock k (1..3): jf, xk and the folded key/value operands of its vertex<-joint
// cross-attention.  Depends only on the joints and the AdaLN parameters, not on the vertex stream.
int joint_prep(pmce_model* m, int k, const float* joints, int B, DecoderWs& w, hipStream_t stream) {
  const int J = m->J;
  const std::string p = "dec.b" + std::to_string(k) + ".";
  const int ib = (k - 1) * 6, gbs = N_ADA * 128;
  RUN(P_JOINT_EMBED, pmce_joint_embed_f32(joints, m->f(p + "joint_proj.weight"), m->f(p + "joint_proj.bias"),
                                          m->f(p + "joint_pos_embed"), m->f(p + "proj_j2v_dim.weight"),
                                          m->f(p + "proj_j2v_dim.bias"), m->f(p + "j2v_K_embed"), w.JF[k - 1], w.XK[k - 1], B,
                                          J, stream));
  RUN(P_CA_FOLD, pmce_ca_fold_f32(w.XK[k - 1], w.JF[k - 1], w.GB, gbs, ib + 0, ib + 1, ib + 2, m->f(p + "vca.wq.weight"),
                                  m->f(p + "vca.wq.bias"), m->f(p + "vca.wk.weight"), m->f(p + "vca.wk.bias"),
                                  m->f(p + "vca.wv.weight"), m->f(p + "vca.wv.bias"), m->f(p + "vca.proj.weight"),
                                  w.KF[k - 1], w.S0[k - 1], w.VF[k - 1], B, J, stream));
  return PMCE_OK;
}

// joint stream of CoevoBlock 3 (the only live one, CoevoDecoder.py:235-237): needs the block's INPUT vertices.
int joint_branch(pmce_model* m, const float* joints, const float* vt_in, float* cam_pose, int B, DecoderWs& w,
                 hipStream_t stream) {
  const int J = m->J, gbs = N_ADA * 128;
  const std::string p = "dec.b3.";
  RUN(P_TOKENS_KV, pmce_tokens_kv_f32(nullptr, nullptr, vt_in, m->f(p + "vertx_proj.weight"), m->f(p + "Ev"),
                                      m->f(p + "proj_v2j_dim.weight"), m->f(p + "Ek"), w.GB, gbs, 19, 20,
                                      m->f(p + "jca.wk.weight"), m->f(p + "jca.wk.bias"), m->f(p + "jca.wv.weight"),
                                      m->f(p + "jca.wv.bias"), w.KVJ, B, stream));
  const float* wp[18] = {m->f(p + "jca.wq.weight"),      m->f(p + "jca.wq.bias"),      m->f(p + "jca.proj.weight"),
                         m->f(p + "jca.proj.bias"),      m->f(p + "jca.mlp.fc1.weight"), m->f(p + "jca.mlp.fc1.bias"),
                         m->f(p + "jca.mlp.fc2.weight"), m->f(p + "jca.mlp.fc2.bias"), m->f(p + "jsa.qkv.weight"),
                         m->f(p + "jsa.qkv.bias"),       m->f(p + "jsa.proj.weight"),  m->f(p + "jsa.proj.bias"),
                         m->f(p + "jsa.mlp.fc1.weight"), m->f(p + "jsa.mlp.fc1.bias"), m->f(p + "jsa.mlp.fc2.weight"),
                         m->f(p + "jsa.mlp.fc2.bias"),   m->f(p + "jcoor.weight"),     m->f(p + "jcoor.bias")};
  const int inst[4] = {18, 21, 22, 23};
  RUN(P_JOINT_STREAM, pmce_joint_stream_f32(w.JF[2], m->f(p + "j_Q_embed"), w.KVJ, w.GB, gbs, wp, inst, joints, nullptr,
                                            cam_pose, B, J, 3, stream));
  return PMCE_OK;
}

// joint/vertex branch of Pose2Mesh.forward (needs the joints and the outputs of gru_part).  With a side stream the
// joint-side work of blocks 2-3 and the joint stream of block 3 run beside the vertex stream (they are short,
// latency-bound kernels); side == nullptr runs everything in order on `stream`.
int coevo_part(pmce_model* m, const float* joints, float* cam_pose, float* cam_mesh, int B, DecoderWs& w,
               hipStream_t stream, hipStream_t side) {
  const int J = m->J;
  const float* g = w.Y1 + (long long)8 * B * 2 * GH;
  const int gbs = N_ADA * 128;
  if (side) {
    (void)hipEventRecord(m->ev_a, stream);  // joints and AdaLN parameters are ready
    (void)hipStreamWaitEvent(side, m->ev_a, 0);
    PMCE_TRY(joint_prep(m, 2, joints, B, w, side));
    PMCE_TRY(joint_prep(m, 3, joints, B, w, side));
    (void)hipEventRecord(m->ev_b, side);
  }
  // ---- vertex init (CoevoDecoder.py:232) ----
  RUN(P_GATHER, pmce_vertex_init_gather_f32(joints, m->i32("dec.vj_relation"), w.VT[0], B, J, stream));
  PMCE_TRY(joint_prep(m, 1, joints, B, w, stream));
  float* vt_cur = w.VT[0];
  for (int k = 1; k <= 3; ++k) {
    const std::string p = "dec.b" + std::to_string(k) + ".";
    float* vt_next = w.VT[k % 3];
    const int ib = (k - 1) * 6;  // AdaLN instances: vca.normq,normk,normv,norm2, vsa.norm1,norm2
    if (k > 1) {
      if (side) {
        if (k == 2) (void)hipStreamWaitEvent(stream, m->ev_b, 0);
      } else {
        PMCE_TRY(joint_prep(m, k, joints, B, w, stream));
      }
    }
    if (k == 3) {
      if (side) {
        (void)hipEventRecord(m->ev_c, stream);  // block-3 input vertices are ready
        (void)hipStreamWaitEvent(side, m->ev_c, 0);
        PMCE_TRY(joint_branch(m, joints, vt_cur, cam_pose, B, w, side));
        (void)hipEventRecord(m->ev_d, side);
      }
    }
    RUN(P_VERTEX_CA, pmce_vertex_ca_f32(nullptr, vt_cur, m->f(p + "vertx_proj.weight"), m->f(p + "Eq"), w.KF[k - 1],
                                        w.S0[k - 1], w.VF[k - 1], m->f(p + "vca.proj.bias"), w.F1, B, J, stream));
    RUN(P_ADALN_MLP, pmce_adaln_mlp_f32(w.F1, w.GB, gbs, ib + 3, m->f(p + "vca.mlp.fc1.weight"), m->f(p + "vca.mlp.fc1.bias"),
                                        m->f(p + "vca.mlp.fc2.weight"), m->f(p + "vca.mlp.fc2.bias"), w.F2, nullptr, nullptr,
                                        nullptr, nullptr, B, stream));
    RUN(P_ADALN_QKV, pmce_adaln_qkv_f32(w.F2, w.GB, gbs, ib + 4, m->f(p + "vsa.qkv.weight"), m->f(p + "vsa.qkv.bias"), w.QKV, B,
                                        stream));
    RUN(P_VERTEX_SA, pmce_vertex_sa_f32(w.F2, w.QKV, m->f(p + "vsa.proj.weight"), m->f(p + "vsa.proj.bias"), w.F1, B, stream));
    RUN(P_ADALN_MLP, pmce_adaln_mlp_f32(w.F1, w.GB, gbs, ib + 5, m->f(p + "vsa.mlp.fc1.weight"), m->f(p + "vsa.mlp.fc1.bias"),
                                        m->f(p + "vsa.mlp.fc2.weight"), m->f(p + "vsa.mlp.fc2.bias"), nullptr,
                                        m->f(p + "vcoor.weight"), m->f(p + "vcoor.bias"), vt_cur, vt_next, B, stream));
    if (k == 3 && !side) PMCE_TRY(joint_branch(m, joints, vt_cur, cam_pose, B, w, stream));
    vt_cur = vt_next;
  }
  // ---- 431 -> 6890 upsample conv + 3 residual Linear(2048->6890) as ONE product (CoevoDecoder.py:238-244) ----
  RUN(P_FINAL_OP, pmce_build_final_operand_f32(g, vt_cur, w.FA, B, FINAL_K, stream));
  RUN(P_GEMM_FINAL, gemm(w.FA, m->f("dec.final.weight"), m->f("dec.final.bias"), nullptr, cam_mesh, B, NVF * 3, FINAL_K,
                         FINAL_K, NVF * 3, 0, stream));
  if (side) (void)hipStreamWaitEvent(stream, m->ev_d, 0);  // cam_pose is written by the side stream
  return PMCE_OK;
}

// second stream + fork/join events, created on first use
// pmce_model_wait_lifter: the point of a forward after which only its decoder is left
void mark_lifter_done(pmce_model* m, hipStream_t stream) {
  if (!m->ev_lifter) (void)hipEventCreateWithFlags(&m->ev_lifter, hipEventDisableTiming);
  if (m->ev_lifter) (void)hipEventRecord(m->ev_lifter, stream);
}

int ensure_side(pmce_model* m) {
  if (m->side) return PMCE_OK;
  int lo = 0, hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
  bool ok = hipStreamCreateWithPriority(&m->side, hipStreamNonBlocking, hi) == hipSuccess;
  for (hipEvent_t* e : {&m->ev_fork, &m->ev_join, &m->ev_a, &m->ev_b, &m->ev_c, &m->ev_d})
    ok = ok && hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    pmce_set_error("could not create the side stream/events: %s", hipGetErrorString(hipGetLastError()));
    return PMCE_ERR_LAUNCH;
  }
  return PMCE_OK;
}

}  // namespace

// ============================================================================================================
extern "C" {

int pmce_model_create(int num_joint, int embed_dim, int depth, pmce_model** out) {
  PMCE_REQUIRE(out, "model_create: null out");
  PMCE_REQUIRE(num_joint >= 17 && num_joint <= 32, "model_create: num_joint must be in 17..32 (vj_relation indexes 0..16)");
  PMCE_REQUIRE(embed_dim == 256 || embed_dim == 512, "model_create: embed_dim must be 256 or 512");
  PMCE_REQUIRE(depth >= 1 && depth <= 8, "model_create: depth must be in 1..8");
  pmce_model* m = new pmce_model();
  m->J = num_joint;
  m->C = embed_dim;
  m->depth = depth;
  m->concurrent = getenv("PMCE_SINGLE_STREAM") == nullptr;
  build_names(m);
  *out = m;
  return PMCE_OK;
}

void pmce_model_destroy(pmce_model* m) {
  if (!m) return;
  for (auto& e : m->pool) {
    (void)hipEventDestroy(e.a);
    (void)hipEventDestroy(e.b);
  }
  for (auto& e : m->pending) {
    (void)hipEventDestroy(e.a);
    (void)hipEventDestroy(e.b);
  }
  for (hipEvent_t e : {m->ev_fork, m->ev_join, m->ev_a, m->ev_b, m->ev_c, m->ev_d, m->ev_lifter})
    if (e) (void)hipEventDestroy(e);
  if (m->side) (void)hipStreamDestroy(m->side);
  delete m;
}

int pmce_model_tensor_count(const pmce_model* m) { return m ? (int)m->names.size() : 0; }
const char* pmce_model_tensor_name(const pmce_model* m, int i) {
  if (!m || i < 0 || i >= (int)m->names.size()) return nullptr;
  return m->names[i].c_str();
}

int pmce_model_set_tensor(pmce_model* m, const char* name, const void* dev_ptr) {
  PMCE_REQUIRE(m && name && dev_ptr, "model_set_tensor: null argument");
  const std::string n(name);
  if (n == "jreg.indptr") { m->jr_indptr = static_cast<const int*>(dev_ptr); return PMCE_OK; }
  if (n == "jreg.indices") { m->jr_indices = static_cast<const int*>(dev_ptr); return PMCE_OK; }
  if (n == "jreg.data") { m->jr_data = static_cast<const float*>(dev_ptr); return PMCE_OK; }
  bool known = false;
  for (auto& s : m->names)
    if (s == n) { known = true; break; }
  PMCE_REQUIRE(known, "model_set_tensor: unknown tensor '%s'", name);
  PMCE_REQUIRE((reinterpret_cast<uintptr_t>(dev_ptr) & 15) == 0, "model_set_tensor: '%s' is not 16-byte aligned", name);
  m->ptr[n] = dev_ptr;
  m->finalized = false;
  return PMCE_OK;
}

int pmce_model_set_regressor_rows(pmce_model* m, int rows) {
  PMCE_REQUIRE(m && rows > 0 && rows <= 32, "model_set_regressor_rows: rows must be in 1..32");
  m->jr_rows = rows;
  return PMCE_OK;
}

int pmce_model_finalize(pmce_model* m) {
  PMCE_REQUIRE(m, "model_finalize: null model");
  // a model may carry only the lifter (LiftTester path, lib/core/base.py:56,357) or only the decoder
  bool any_l = false, all_l = true, any_d = false, all_d = true;
  const char* missing_l = nullptr;
  const char* missing_d = nullptr;
  for (auto& s : m->names) {
    const bool have = m->ptr.count(s) != 0;
    if (s.rfind("lifter.", 0) == 0) {
      any_l |= have;
      if (!have) { all_l = false; if (!missing_l) missing_l = s.c_str(); }
    } else {
      any_d |= have;
      if (!have) { all_d = false; if (!missing_d) missing_d = s.c_str(); }
    }
  }
  PMCE_REQUIRE(any_l || any_d, "model_finalize: no tensors registered");
  PMCE_REQUIRE(!any_l || all_l, "model_finalize: lifter tensor '%s' was never registered", missing_l ? missing_l : "?");
  PMCE_REQUIRE(!any_d || all_d, "model_finalize: decoder tensor '%s' was never registered", missing_d ? missing_d : "?");
  m->has_lifter = all_l && any_l;
  m->has_decoder = all_d && any_d;
  m->finalized = true;
  return PMCE_OK;
}

size_t pmce_model_workspace_bytes(const pmce_model* m, int batch) {
  if (!m || batch <= 0) return 0;
  Carver c(nullptr, ~(size_t)0);
  LifterWs lw;
  DecoderWs dw;
  carve_lifter(c, m, batch, lw);
  carve_decoder(c, m, batch, dw);
  return c.off + 256;
}

long long pmce_model_workspace_offset(const pmce_model* m, int batch, const char* name) {
  if (!m || batch <= 0 || !name) return -1;
  Carver c(nullptr, ~(size_t)0);
  LifterWs lw;
  DecoderWs dw;
  carve_lifter(c, m, batch, lw);
  carve_decoder(c, m, batch, dw);
  const std::string n(name);
  const float* p = nullptr;
  if (n == "X") p = lw.X;                                                  // lifter tokens [B,16,J,C]
  else if (n == "Y0") p = dw.Y0;                                           // GRU layer-0 output [16,B,2048]
  else if (n == "g") p = dw.Y1 + (long long)8 * batch * 2 * GH;            // y[8] [B,2048]
  else if (n == "GB") p = dw.GB;                                           // AdaLN gamma|beta [B,24*128]
  else if (n == "VT0") p = dw.VT[0];                                       // after a forward: v3
  else if (n == "VT1") p = dw.VT[1];                                       // v1
  else if (n == "VT2") p = dw.VT[2];                                       // v2
  else if (n == "F1") p = dw.F1;
  else if (n == "F2") p = dw.F2;
  else if (n == "JM") p = dw.JM;
  else return -1;
  return (long long)(reinterpret_cast<const char*>(p) - static_cast<const char*>(nullptr));
}

static int check_ws(pmce_model* m, int batch, void* ws, size_t ws_bytes) {
  PMCE_REQUIRE(m && m->finalized, "model not finalized (call pmce_model_finalize after registering all tensors)");
  PMCE_REQUIRE(batch > 0, "batch must be positive");
  PMCE_REQUIRE(ws && (reinterpret_cast<uintptr_t>(ws) & 255) == 0, "workspace must be non-null and 256-byte aligned");
  if (ws_bytes < pmce_model_workspace_bytes(m, batch)) {
    pmce_set_error("workspace too small: %zu < %zu bytes for batch %d", ws_bytes, pmce_model_workspace_bytes(m, batch), batch);
    return PMCE_ERR_WORKSPACE;
  }
  return PMCE_OK;
}

int pmce_lifter_forward(pmce_model* m, const float* pose2d, const float* img_feat, float* pose3d, int batch, void* ws,
                        size_t ws_bytes, pmce_stream_t stream) {
  PMCE_TRY(check_ws(m, batch, ws, ws_bytes));
  PMCE_REQUIRE(m->has_lifter, "lifter_forward: lifter tensors not registered");
  PMCE_REQUIRE(pose2d && img_feat && pose3d, "lifter_forward: null pointer");
  Carver c(ws, ws_bytes);
  LifterWs lw;
  carve_lifter(c, m, batch, lw);
  return lifter_impl(m, pose2d, img_feat, pose3d, batch, lw, stream);
}

int pmce_decoder_forward(pmce_model* m, const float* joints, const float* img_feat, float* cam_pose, float* cam_mesh,
                         int batch, void* ws, size_t ws_bytes, pmce_stream_t stream) {
  PMCE_TRY(check_ws(m, batch, ws, ws_bytes));
  PMCE_REQUIRE(m->has_decoder, "decoder_forward: decoder tensors not registered");
  PMCE_REQUIRE(joints && img_feat && cam_pose && cam_mesh, "decoder_forward: null pointer");
  Carver c(ws, ws_bytes);
  LifterWs lw;
  DecoderWs dw;
  carve_lifter(c, m, batch, lw);
  carve_decoder(c, m, batch, dw);
  PMCE_TRY(gru_part(m, img_feat, batch, dw, stream));
  if (m->concurrent) PMCE_TRY(ensure_side(m));
  return coevo_part(m, joints, cam_pose, cam_mesh, batch, dw, stream, m->concurrent ? m->side : nullptr);
}

int pmce_forward(pmce_model* m, const float* pose2d, const float* img_feat, float* cam_mesh, float* cam_pose,
                 float* pose3d, float* pred_pose, int batch, void* ws, size_t ws_bytes, pmce_stream_t stream) {
  PMCE_TRY(check_ws(m, batch, ws, ws_bytes));
  PMCE_REQUIRE(m->has_lifter && m->has_decoder, "forward: needs both lifter and decoder tensors");
  PMCE_REQUIRE(pose2d && img_feat && cam_mesh && cam_pose && pose3d, "forward: null pointer");
  Carver c(ws, ws_bytes);
  LifterWs lw;
  DecoderWs dw;
  carve_lifter(c, m, batch, lw);
  carve_decoder(c, m, batch, dw);
  // Fork: the GRU / AdaLN-parameter branch depends only on img_feat; it runs on a second (high-priority) stream
  // under the pose lifter, whose long matrix-core kernels leave the gaps its 25 short dependent steps need.
  // pmce_model_set_concurrency(m, 0) (or PMCE_SINGLE_STREAM=1 at create time) keeps everything on one stream.
  const bool single = !m->concurrent;
  if (!single) PMCE_TRY(ensure_side(m));
  if (!single) {
    (void)hipEventRecord(m->ev_fork, stream);
    (void)hipStreamWaitEvent(m->side, m->ev_fork, 0);
    PMCE_TRY(gru_part(m, img_feat, batch, dw, m->side));
    (void)hipEventRecord(m->ev_join, m->side);
  } else {
    PMCE_TRY(gru_part(m, img_feat, batch, dw, stream));
  }
  PMCE_TRY(lifter_impl(m, pose2d, img_feat, pose3d, batch, lw, stream));
  // pose3d.reshape(-1, J, 3) / 1000  (PMCE.py:17-18)
  RUN(P_MISC, pmce_div_scalar_f32(pose3d, dw.JM, (long long)batch * m->J * 3, 1000.0f, stream));
  mark_lifter_done(m, stream);
  if (!single) (void)hipStreamWaitEvent(stream, m->ev_join, 0);  // join
  PMCE_TRY(coevo_part(m, dw.JM, cam_pose, cam_mesh, batch, dw, stream, single ? nullptr : m->side));
  if (pred_pose) {
    PMCE_REQUIRE(m->jr_indptr && m->jr_indices && m->jr_data && m->jr_rows > 0,
                 "forward: pred_pose requested but no J_regressor registered (jreg.indptr/indices/data + rows)");
    RUN(P_JREG, pmce_j_regress_f32(cam_mesh, m->jr_indptr, m->jr_indices, m->jr_data, pred_pose, batch, m->jr_rows, NVF,
                                   1000.0f, stream));
  }
  return PMCE_OK;
}

// ---- streaming (stride-1 windows over one long sequence) -------------------------------------------
int pmce_window_tokens_f32(const float* x0, const int* win, const float* tpos, const float* w2, const float* b2, float eps2,
                           float* X, float* XN, int W, int L, int T, int J, int C, hipStream_t stream);
int pmce_window_rows_f32(const float* src, const int* win, float* dst, int W, int L, int T, int ncols, hipStream_t stream);

int pmce_stream_precompute(pmce_model* m, const float* pose2d_frames, const float* feat_frames, int L, float* x0, float* gi0,
                           void* ws, size_t ws_bytes, pmce_stream_t stream) {
  PMCE_REQUIRE(L > 0, "stream_precompute: L must be positive");
  const int bf = (L + T - 1) / T;  // the per-frame pass needs the workspace of ceil(L/16) clips
  PMCE_TRY(check_ws(m, bf, ws, ws_bytes));
  PMCE_REQUIRE(m->has_lifter && m->has_decoder, "stream_precompute: needs both lifter and decoder tensors");
  PMCE_REQUIRE(pose2d_frames && feat_frames && x0 && gi0, "stream_precompute: null pointer");
  Carver c(ws, ws_bytes);
  LifterWs lw;
  carve_lifter(c, m, bf, lw);
  // window-independent lifter work: embedding + SpatialBlocks[0] + norm_s, once per frame (PoseEstimation.py:78-85)
  PMCE_TRY(lifter_frames(m, pose2d_frames, feat_frames, L, lw, stream));
  RUN(P_LN, pmce_ln_chain_f32(lw.X, (long long)L * m->J, m->C, m->f("lifter.norm_s.weight"), m->f("lifter.norm_s.bias"), 1e-6f,
                              nullptr, 1, 1, x0, nullptr, nullptr, 0.f, nullptr, stream));
  // window-independent GRU work: layer-0 input projections of both directions, once per frame (CoevoDecoder.py:216-221)
  RUN(P_GEMM_GRU_IN, gemm(feat_frames, m->f("dec.gru.w_ih_l0"), m->f("dec.gru.b_ih_l0"), nullptr, gi0, L, 6 * GH, F, F, 6 * GH,
                          0, stream));
  return PMCE_OK;
}

int pmce_stream_forward(pmce_model* m, const float* x0, const float* gi0, const int* win, int W, int L, float* cam_mesh,
                        float* cam_pose, float* pose3d, float* pred_pose, void* ws, size_t ws_bytes, pmce_stream_t stream) {
  PMCE_TRY(check_ws(m, W, ws, ws_bytes));
  PMCE_REQUIRE(m->has_lifter && m->has_decoder, "stream_forward: needs both lifter and decoder tensors");
  PMCE_REQUIRE(x0 && gi0 && win && cam_mesh && cam_pose && pose3d && L > 0, "stream_forward: null pointer");
  Carver c(ws, ws_bytes);
  LifterWs lw;
  DecoderWs dw;
  carve_lifter(c, m, W, lw);
  carve_decoder(c, m, W, dw);
  const bool single = !m->concurrent;
  if (!single) PMCE_TRY(ensure_side(m));
  hipStream_t gs = single ? stream : m->side;
  if (!single) {
    (void)hipEventRecord(m->ev_fork, stream);
    (void)hipStreamWaitEvent(m->side, m->ev_fork, 0);
  }
  {
    hipStream_t stream_save = stream;
    stream = gs;  // RUN() launches on `stream`
    RUN(P_MISC, pmce_window_rows_f32(gi0, win, dw.GI0, W, L, T, 6 * GH, stream));
    stream = stream_save;
  }
  PMCE_TRY(gru_rest(m, W, dw, gs));
  if (!single) (void)hipEventRecord(m->ev_join, m->side);
  RUN(P_LN, pmce_window_tokens_f32(x0, win, m->f("lifter.temporal_pos_embed"), m->f(blk("Temporal", 0, "norm1.weight")),
                                   m->f(blk("Temporal", 0, "norm1.bias")), 1e-6f, lw.X, lw.XN, W, L, T, m->J, m->C, stream));
  PMCE_TRY(lifter_rest(m, pose3d, W, lw, stream));
  RUN(P_MISC, pmce_div_scalar_f32(pose3d, dw.JM, (long long)W * m->J * 3, 1000.0f, stream));
  mark_lifter_done(m, stream);
  if (!single) (void)hipStreamWaitEvent(stream, m->ev_join, 0);
  PMCE_TRY(coevo_part(m, dw.JM, cam_pose, cam_mesh, W, dw, stream, single ? nullptr : m->side));
  if (pred_pose) {
    PMCE_REQUIRE(m->jr_indptr && m->jr_indices && m->jr_data && m->jr_rows > 0, "stream_forward: no J_regressor registered");
    RUN(P_JREG, pmce_j_regress_f32(cam_mesh, m->jr_indptr, m->jr_indices, m->jr_data, pred_pose, W, m->jr_rows, NVF, 1000.0f,
                                   stream));
  }
  return PMCE_OK;
}

int pmce_model_wait_lifter(pmce_model* m, pmce_stream_t stream) {
  PMCE_REQUIRE(m, "model_wait_lifter: null model");
  if (m->ev_lifter && hipStreamWaitEvent(stream, m->ev_lifter, 0) != hipSuccess) {
    pmce_set_error("model_wait_lifter: %s", hipGetErrorString(hipGetLastError()));
    return PMCE_ERR_LAUNCH;
  }
  return PMCE_OK;  // no forward has run on m yet: nothing to wait for
}

int pmce_model_set_concurrency(pmce_model* m, int enable) {
  PMCE_REQUIRE(m, "model_set_concurrency: null model");
  m->concurrent = enable != 0;
  return PMCE_OK;
}

int pmce_model_profile(pmce_model* m, int enable) {
  PMCE_REQUIRE(m, "model_profile: null model");
  m->prof = enable != 0;
  if (enable) {
    for (int i = 0; i < P_COUNT; ++i) {
      m->prof_ms[i] = 0;
      m->prof_n[i] = 0;
    }
  }
  return PMCE_OK;
}

int pmce_model_profile_read(pmce_model* m, int i, const char** name, double* ms, long long* launches) {
  if (!m) return 0;
  if (!m->pending.empty()) {
    for (auto& e : m->pending) {
      (void)hipEventSynchronize(e.b);
      float t = 0.f;
      if (hipEventElapsedTime(&t, e.a, e.b) == hipSuccess) {
        m->prof_ms[e.cls] += t;
        m->prof_n[e.cls] += 1;
      }
      m->pool.push_back(e);
    }
    m->pending.clear();
  }
  if (i >= 0 && i < P_COUNT) {
    if (name) *name = kProfNames[i];
    if (ms) *ms = m->prof_ms[i];
    if (launches) *launches = m->prof_n[i];
  }
  return P_COUNT;
}

}  // extern "C"
