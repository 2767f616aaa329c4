// Host-side orchestration of the PMCE hot path: a descriptor of registered device pointers plus the launch
// sequences of GraphormerNet.forward (reference lib/models/PoseEstimation.py:95-115), Pose2Mesh.forward
// (lib/models/CoevoDecoder.py:226-246), PMCE.forward (lib/models/PMCE.py:15-20) and the caller's J_regressor
// projection (lib/core/base.py:223-225).  No device memory is allocated here: weights and workspace belong to
// the caller.  Launches go to the caller's stream and - for the branches that are independent of it - to one
// internally created side stream, forked and joined with events (no host synchronisation: a forward is
// hipGraph-capturable).  Every tensor pointer is resolved into a plain struct at pmce_model_finalize, so a launch
// sequence does no string or map work (it matters at batch 1, the reference demo's batch: main/run_demo.py:332).
#include <hip/hip_runtime.h>

#include <stdlib.h>

#include <memory>
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
  P_GATHER, P_JOINT_EMBED, P_CA_FOLD, P_VERTEX_CA, P_VERTEX_CA_MLP, P_ADALN_MLP, P_ADALN_QKV, P_VERTEX_SA, P_TOKENS_KV, P_JOINT_STREAM,
  P_FINAL_OP, P_JREG, P_MISC, P_COUNT
};
const char* kProfNames[P_COUNT] = {
    "gemm_lifter", "gemm_gru_in", "gru_step", "gemm_ada", "gemm_final", "ln_chain", "seq_attention", "embed_tokens",
    "lifter_head", "vertex_init_gather", "joint_embed", "ca_fold", "vertex_ca", "vertex_ca_mlp", "adaln_mlp", "adaln_qkv",
    "vertex_sa", "tokens_kv", "joint_stream", "build_final_operand", "j_regress", "misc"};

struct Ev {
  hipEvent_t a, b;
  int cls;
};


// ---- resolved tensor pointers (filled by pmce_model_finalize from the registered names) -----------------------
struct LifterBlockW {
  const float *norm1_w, *norm1_b, *qkv_w, *qkv_b, *proj_w, *proj_b, *norm2_w, *norm2_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
};
// the lifter's Linear weights as f16 (hi, lo) planes + scale (pmce_gemm_pack_split_f16): made by pmce_model_finalize in
// memory the model owns when the lifter's products run in the three-product f16 form
struct SplitW {
  const float* wp = nullptr;
  const float* scale = nullptr;
};
struct LifterBlockSplit {
  SplitW qkv, proj, fc1, fc2;
};
struct VertexBlockW {  // vertex side of CoevoBlock k (+ the joint-side preparation every block needs)
  const float *joint_proj_w, *joint_proj_b, *joint_pos, *j2v_w, *j2v_b, *j2v_K, *vertx_proj_w, *Eq;
  const float *vca_wq_w, *vca_wq_b, *vca_wk_w, *vca_wk_b, *vca_wv_w, *vca_wv_b, *vca_proj_w, *vca_proj_b;
  const float *vca_fc1_w, *vca_fc1_b, *vca_fc2_w, *vca_fc2_b;
  const float *vsa_qkv_w, *vsa_qkv_b, *vsa_proj_w, *vsa_proj_b, *vsa_fc1_w, *vsa_fc1_b, *vsa_fc2_w, *vsa_fc2_b;
  const float *vcoor_w, *vcoor_b;
};
struct JointBlockW {  // joint stream of CoevoBlock 3 (the only live one)
  const float *Ev, *v2j_w, *Ek, *j_Q, *jca_wk_w, *jca_wk_b, *jca_wv_w, *jca_wv_b;
  const float* stream[18];  // joint_stream's weight table (order: see pmce_joint_stream_f32)
};
struct Weights {
  const float *je_w, *je_b, *ie_w, *ie_b, *spos, *tpos;
  LifterBlockW blk[2][8];  // [0 spatial | 1 temporal][depth]
  const float *ns_w, *ns_b, *nt_w, *nt_b, *reg0_w, *reg0_b, *reg1_w, *reg1_b, *fus_w, *fus_b;
  const int* vj;
  const float *wih0, *bih0, *whh0, *bhh0, *wih1, *bih1, *whh1, *bhh1, *ada_w, *ada_b;
  VertexBlockW vb[3];
  JointBlockW jb;
  const float *final_w, *final_b;
};
struct Slot {
  std::string name;
  const void** dst;
};

}  // namespace

struct pmce_model {
  int J, C, depth;
  std::vector<std::string> names;
  std::vector<Slot> slots;  // names[i] -> field of `w`
  std::unordered_map<std::string, const void*> ptr;
  Weights w = {};
  bool finalized = false;
  bool has_lifter = false, has_decoder = false;
  // second stream for the image-feature branch (created on first use, destroyed with the model)
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_a = nullptr, ev_b = nullptr, ev_c = nullptr, ev_d = nullptr;
  hipEvent_t ev_lifter = nullptr;  // recorded by pmce_forward when its pose lifter is enqueued (pmce_model_wait_lifter)
  bool concurrent = true;  // pmce_model_set_concurrency
  // the large products on the f16 matrix pipe (three-product split, fp32 accuracy) instead of the fp32 one
  bool split_gemm = true;  // pmce_model_set_gemm_mode / PMCE_SPLIT_F16=0 at create
  std::shared_ptr<float> split_arena;  // every packed weight + its per-row 2^-s; shared by handles cloned onto the same weights.
                                       // Caller memory (pmce_model_set_split_arena: no-op deleter) or, without one, hipMalloc'ed.
  float* caller_arena = nullptr;       // set by pmce_model_set_split_arena, consumed by the next (re)build of the planes
  size_t caller_arena_bytes = 0;
  bool split_adopted = false;          // the planes came from another handle (pmce_model_share_split_weights): finalize keeps them
  LifterBlockSplit sblk[2][8];
  SplitW s_ie, s_wih0, s_wih1, s_whh0, s_whh1, s_ada, s_final;
  // Below this many clips per call the products stay on the fp32 pipe (PMCE_SPLIT_MIN_BATCH at create).  Default 1 = never: with
  // the two-stream schedule in both modes the f16 form is faster at every batch size (B = 1: 1.57 ms against 1.85 ms at C = 512;
  // scripts/microbench/small_batch_modes.py).  With PMCE_SPLIT_OVERLAP=0 a threshold near 48 pays: a small batch is bound by its
  // 25 dependent GRU launches, which only the second stream hides.
  int split_min_batch = 1;
  // Two streams also in split mode.  f16 matrix instructions disturb packed-fp32 arithmetic of co-resident waves on MI355X (DESIGN
  // 3.4); the library therefore contains no packed-fp32 instruction at all (build.py), which makes its kernels safe next to each
  // other.  PMCE_SPLIT_OVERLAP=0 at create restores the strictly serial schedule of the split mode (diagnostic).
  bool split_overlap = true;
  const float* ffn_img[3][2] = {};     // per vertex block: the LDS images of the two FFNs' f16 form (vca, vsa), in the split arena
  const float* qkv_img[3] = {};        // per vertex block: the self-attention qkv weight's f16 form (vertex_sab)
  const float* tkv_img = nullptr;      // the joint<-vertex direction's three 64 x 64 weights (tokens_kv), block 3
  bool split_now = false;  // decision for the call in progress (set by check_ws, the first thing every entry point does)
  // Sticky "a product of this model produced a non-finite value" word: 4 bytes of pinned host memory the device can write
  // (hipHostMalloc, mapped), so that reading it costs no synchronisation.  Set by the split-f16 products' epilogues (an activation
  // beyond f16's 65504 turns into inf / nan there, and so does fp32 overflow), checked by every entry point BEFORE it launches.
  std::shared_ptr<unsigned> oflow;  // shared by handles cloned onto the same weights (pipeline lanes): one model, one flag
  bool strict_overflow = false;     // pmce_model_set_overflow_policy: refuse further calls while the word is set
  unsigned long long* clk = nullptr;  // pmce_model_set_clock_probe: two device words (null = off)
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

// fork/join primitives with checked return codes: a failed record/wait would silently race the two streams
int ev_record(hipEvent_t e, hipStream_t s, const char* what) {
  const hipError_t rc = hipEventRecord(e, s);
  if (rc != hipSuccess) {
    pmce_set_error("%s: hipEventRecord failed: %s", what, hipGetErrorString(rc));
    return PMCE_ERR_LAUNCH;
  }
  return PMCE_OK;
}
int ev_wait(hipStream_t s, hipEvent_t e, const char* what) {
  const hipError_t rc = hipStreamWaitEvent(s, e, 0);
  if (rc != hipSuccess) {
    pmce_set_error("%s: hipStreamWaitEvent failed: %s", what, hipGetErrorString(rc));
    return PMCE_ERR_LAUNCH;
  }
  return PMCE_OK;
}

#define RUN(cls, call)             \
  do {                             \
    ProfScope _ps(m, cls, stream); \
    PMCE_TRY(call);                \
  } while (0)

// Tensor names (what pmce_model_tensor_name enumerates) and, for each, the field of m->w it resolves into.
void build_names(pmce_model* m) {
  Weights& w = m->w;
  auto add = [&](const std::string& n, const float*& field) {
    m->names.push_back(n);
    m->slots.push_back({n, reinterpret_cast<const void**>(&field)});
  };
  add("lifter.joint_embed.weight", w.je_w);
  add("lifter.joint_embed.bias", w.je_b);
  add("lifter.imgfeat_embed.weight", w.ie_w);
  add("lifter.imgfeat_embed.bias", w.ie_b);
  add("lifter.spatial_pos_embed", w.spos);
  add("lifter.temporal_pos_embed", w.tpos);
  for (int kind = 0; kind < 2; ++kind)
    for (int i = 0; i < m->depth; ++i) {
      const std::string p = std::string("lifter.") + (kind == 0 ? "Spatial" : "Temporal") + "Blocks." + std::to_string(i) + ".";
      LifterBlockW& b = w.blk[kind][i];
      add(p + "norm1.weight", b.norm1_w);
      add(p + "norm1.bias", b.norm1_b);
      add(p + "attn.qkv.weight", b.qkv_w);
      add(p + "attn.qkv.bias", b.qkv_b);
      add(p + "attn.proj.weight", b.proj_w);
      add(p + "attn.proj.bias", b.proj_b);
      add(p + "norm2.weight", b.norm2_w);
      add(p + "norm2.bias", b.norm2_b);
      add(p + "mlp.fc1.weight", b.fc1_w);
      add(p + "mlp.fc1.bias", b.fc1_b);
      add(p + "mlp.fc2.weight", b.fc2_w);
      add(p + "mlp.fc2.bias", b.fc2_b);
    }
  add("lifter.norm_s.weight", w.ns_w);
  add("lifter.norm_s.bias", w.ns_b);
  add("lifter.norm_t.weight", w.nt_w);
  add("lifter.norm_t.bias", w.nt_b);
  add("lifter.regression.0.weight", w.reg0_w);
  add("lifter.regression.0.bias", w.reg0_b);
  add("lifter.regression.1.weight", w.reg1_w);
  add("lifter.regression.1.bias", w.reg1_b);
  add("lifter.fusion.weight", w.fus_w);
  add("lifter.fusion.bias", w.fus_b);
  m->names.push_back("dec.vj_relation");
  m->slots.push_back({"dec.vj_relation", reinterpret_cast<const void**>(&w.vj)});
  add("dec.gru.w_ih_l0", w.wih0);
  add("dec.gru.b_ih_l0", w.bih0);
  add("dec.gru.w_hh_l0", w.whh0);
  add("dec.gru.b_hh_l0", w.bhh0);
  add("dec.gru.w_ih_l1", w.wih1);
  add("dec.gru.b_ih_l1", w.bih1);
  add("dec.gru.w_hh_l1", w.whh1);
  add("dec.gru.b_hh_l1", w.bhh1);
  add("dec.ada.weight", w.ada_w);
  add("dec.ada.bias", w.ada_b);
  for (int k = 1; k <= 3; ++k) {
    const std::string p = "dec.b" + std::to_string(k) + ".";
    VertexBlockW& v = w.vb[k - 1];
    add(p + "joint_proj.weight", v.joint_proj_w);
    add(p + "joint_proj.bias", v.joint_proj_b);
    add(p + "joint_pos_embed", v.joint_pos);
    add(p + "proj_j2v_dim.weight", v.j2v_w);
    add(p + "proj_j2v_dim.bias", v.j2v_b);
    add(p + "j2v_K_embed", v.j2v_K);
    add(p + "vertx_proj.weight", v.vertx_proj_w);
    add(p + "Eq", v.Eq);
    add(p + "vca.wq.weight", v.vca_wq_w);
    add(p + "vca.wq.bias", v.vca_wq_b);
    add(p + "vca.wk.weight", v.vca_wk_w);
    add(p + "vca.wk.bias", v.vca_wk_b);
    add(p + "vca.wv.weight", v.vca_wv_w);
    add(p + "vca.wv.bias", v.vca_wv_b);
    add(p + "vca.proj.weight", v.vca_proj_w);
    add(p + "vca.proj.bias", v.vca_proj_b);
    add(p + "vca.mlp.fc1.weight", v.vca_fc1_w);
    add(p + "vca.mlp.fc1.bias", v.vca_fc1_b);
    add(p + "vca.mlp.fc2.weight", v.vca_fc2_w);
    add(p + "vca.mlp.fc2.bias", v.vca_fc2_b);
    add(p + "vsa.qkv.weight", v.vsa_qkv_w);
    add(p + "vsa.qkv.bias", v.vsa_qkv_b);
    add(p + "vsa.proj.weight", v.vsa_proj_w);
    add(p + "vsa.proj.bias", v.vsa_proj_b);
    add(p + "vsa.mlp.fc1.weight", v.vsa_fc1_w);
    add(p + "vsa.mlp.fc1.bias", v.vsa_fc1_b);
    add(p + "vsa.mlp.fc2.weight", v.vsa_fc2_w);
    add(p + "vsa.mlp.fc2.bias", v.vsa_fc2_b);
    add(p + "vcoor.weight", v.vcoor_w);
    add(p + "vcoor.bias", v.vcoor_b);
  }
  {
    const std::string p = "dec.b3.";
    JointBlockW& j = w.jb;
    add(p + "Ev", j.Ev);
    add(p + "proj_v2j_dim.weight", j.v2j_w);
    add(p + "Ek", j.Ek);
    add(p + "j_Q_embed", j.j_Q);
    const char* leaves[18] = {"jca.wq.weight",      "jca.wq.bias",      "jca.proj.weight",    "jca.proj.bias",
                              "jca.mlp.fc1.weight", "jca.mlp.fc1.bias", "jca.mlp.fc2.weight", "jca.mlp.fc2.bias",
                              "jsa.qkv.weight",     "jsa.qkv.bias",     "jsa.proj.weight",    "jsa.proj.bias",
                              "jsa.mlp.fc1.weight", "jsa.mlp.fc1.bias", "jsa.mlp.fc2.weight", "jsa.mlp.fc2.bias",
                              "jcoor.weight",       "jcoor.bias"};
    // registration order of round 1 kept (jca.wk/wv sit between wq and proj there)
    add(p + leaves[0], j.stream[0]);
    add(p + leaves[1], j.stream[1]);
    add(p + "jca.wk.weight", j.jca_wk_w);
    add(p + "jca.wk.bias", j.jca_wk_b);
    add(p + "jca.wv.weight", j.jca_wv_w);
    add(p + "jca.wv.bias", j.jca_wv_b);
    for (int i = 2; i < 18; ++i) add(p + leaves[i], j.stream[i]);
  }
  add("dec.final.weight", w.final_w);
  add("dec.final.bias", w.final_b);
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
  float *FS, *FR;  // split mode: the raw image features as row-scaled f16 planes [frames][2048] + 2^e per frame (prep_features)
};
struct DecoderWs {
  float *GI0, *Y0, *Y0P, *GI1, *Y1, *GB, *VT[3], *JF[3], *KF[3], *S0[3], *VF[3], *CAI[3], *F1, *F2, *QKV, *KVJ, *FA, *JM;
};

void carve_lifter(Carver& c, const pmce_model* m, int B, LifterWs& w) {
  const size_t M = (size_t)B * T * m->J, C = m->C;
  w.E = c.take((size_t)B * T * C);
  w.X = c.take(M * C);
  w.XN = c.take(M * C);
  w.QKV = c.take(M * 3 * C);  // also the MLP hidden [M,2C]
  w.AO = c.take(M * C);
  w.FS = c.take((size_t)B * T * F);
  w.FR = c.take((size_t)B * T);
}
void carve_decoder(Carver& c, const pmce_model* m, int B, DecoderWs& w) {
  w.GI0 = c.take((size_t)T * B * 6 * GH);
  w.Y0 = c.take((size_t)T * B * 2 * GH);
  w.Y0P = c.take((size_t)T * B * 2 * GH);  // the same rows pre-split (hi | lo*2^11 f16): the layer-1 projections' A operand in the split-f16 form
  w.GI1 = c.take((size_t)2 * 9 * B * 3 * GH);
  w.Y1 = c.take((size_t)T * B * 2 * GH);
  w.GB = c.take((size_t)B * N_ADA * 128);
  for (int i = 0; i < 3; ++i) w.VT[i] = c.take((size_t)B * NVC * 3);
  for (int i = 0; i < 3; ++i) {
    w.JF[i] = c.take((size_t)B * 32 * D);
    w.KF[i] = c.take((size_t)B * 4096);
    w.S0[i] = c.take((size_t)B * 64);
    w.VF[i] = c.take((size_t)B * 4096);
    w.CAI[i] = c.take((size_t)B * pmce_ca_image_floats());  // the folded operands as f16 planes (vertex_ca_mlp's split form)
  }
  w.F1 = c.take((size_t)B * NVC * D);
  w.F2 = c.take((size_t)B * NVC * D);
  // fp32 QKV [B,431,192] of the two-launch self-attention (fp32 pipe) or the fused kernel's key-tile scratch (pmce_vertex_sab_scratch_floats)
  w.QKV = c.take(std::max((size_t)B * NVC * 3 * D, (size_t)pmce_vertex_sab_scratch_floats(B)));
  w.KVJ = c.take((size_t)B * NVC * 2 * D);
  w.FA = c.take((size_t)B * FINAL_K);
  w.JM = c.take((size_t)B * 32 * 3);
}

int gemm(const float* A, const float* W, const float* bias, const float* R, float* Cc, int M, int N, int K, long long lda,
         long long ldc, int act, hipStream_t s) {
  return pmce_gemm_nt_f32(A, W, bias, R, Cc, M, N, K, lda, K, ldc, act, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, s);
}
// a large product: the f16 three-product form when the model carries the packed weight, the fp32 pipe otherwise
int lgemm(const pmce_model* m, const float* A, const float* W, const SplitW& sw, const float* bias, const float* R, float* Cc,
          int M, int N, int K, long long lda, long long ldc, int act, hipStream_t s, int a_packed = 0, int c_packed = 0) {
  if (m->split_now && sw.wp)
    return pmce_gemm_nt_split_f16_blk(A, nullptr, sw.wp, sw.scale, bias, R, Cc, M, N, K, lda, ldc, act, a_packed, c_packed, 0, 0, 0, s);
  PMCE_REQUIRE(!a_packed && !c_packed, "lgemm: pre-split operands need the split-f16 form");
  return gemm(A, W, bias, R, Cc, M, N, K, lda, ldc, act, s);
}
// In the split-f16 form the producers of the lifter blocks' GEMM operands (LayerNorm -> XN, attention -> AO, fc1 -> Hid) write
// them pre-split (hi | lo*2^11 f16 planes in the bytes of the fp32 row): the products then spend no vector work on splitting.
inline int pk(const pmce_model* m) { return m->split_now ? 1 : 0; }
// the decoder's token-local kernels in the same form
inline int pkf(const pmce_model* m) { return m->split_now ? 1 : 0; }

// ---- GraphormerNet.forward --------------------------------------------------------------------------------
// Everything up to and including SpatialBlocks[0] is PER FRAME (its attention runs over the J joints of one frame,
// PoseEstimation.py:78-85), so it is written over `nframes` frames: B*16 for independent clips, L for a streamed sequence.

// the LayerNorm chain that follows a block (lifter_post_norm): norm_s / norm_t, then the next block's norm1
struct PostNorm { const float *w1, *b1, *w2, *b2; };
PostNorm post_norm_of(const pmce_model* m, int kind, int i) {
  PostNorm pn;
  pn.w1 = kind == 0 ? m->w.ns_w : m->w.nt_w;
  pn.b1 = kind == 0 ? m->w.ns_b : m->w.nt_b;
  pn.w2 = pn.b2 = nullptr;
  if (kind == 0) {
    pn.w2 = m->w.blk[1][i].norm1_w;
    pn.b2 = m->w.blk[1][i].norm1_b;
  } else if (i + 1 < m->depth) {
    pn.w2 = m->w.blk[0][i + 1].norm1_w;
    pn.b2 = m->w.blk[0][i + 1].norm1_b;
  }
  return pn;
}
// C = 256 in split mode: the N = 256 products own whole rows (64 x 256 tiles), so the LayerNorm that follows them runs in their epilogue
// (pmce_gemm_nt_split_f16_ln) instead of a launch of its own that re-reads the row.  (C = 512: a 512-wide tile does not fit)
inline bool ln_in_product(const pmce_model* m) { return m->split_now && m->C == 256; }

// attention + MLP of one block (pre-norm input in w.XN, residual stream in w.X); kind 0 spatial, 1 temporal.  post != nullptr: the
// block's post-norm chain (without a position embedding) is part of this call - fused into fc2 where ln_in_product, a launch otherwise.
int lifter_block_body(pmce_model* m, int kind, int i, long long M, int nframes, int B, LifterWs& w, hipStream_t stream,
                      const PostNorm* post = nullptr) {
  const int J = m->J, C = m->C;
  const LifterBlockW& bw = m->w.blk[kind][i];
  const LifterBlockSplit& sw = m->sblk[kind][i];
  // split mode: the attention runs on the f16 matrix pipe too (seq_attention_mfma.hip; reads the fp32 q, k, v, writes AO pre-split)
  const int N_seq = kind == 0 ? J : T;
  const bool mfma_attn = m->split_now && pmce_seq_attention_split_supported(N_seq, C);
  RUN(P_GEMM_LIFTER, lgemm(m, w.XN, bw.qkv_w, sw.qkv, bw.qkv_b, nullptr, w.QKV,
                          (int)M, 3 * C, C, C, 3 * C, 0, stream, pk(m)));
  if (kind == 0) {  // sequences = frames, tokens j contiguous                      (PoseEstimation.py:78,101)
    if (mfma_attn) RUN(P_SEQ_ATTN, pmce_seq_attention_split_f16(w.QKV, w.AO, nframes, J, C, 0, J, 0, 1, stream));
    else RUN(P_SEQ_ATTN, pmce_seq_attention_ex_f32(w.QKV, w.AO, nframes, J, C, 0, J, 0, 1, pk(m), stream));
  } else {  // sequences = (b,j), tokens t at stride J                              (PoseEstimation.py:87,104)
    if (mfma_attn) RUN(P_SEQ_ATTN, pmce_seq_attention_split_f16(w.QKV, w.AO, B * J, T, C, J, 1, (long long)T * J, J, stream));
    else RUN(P_SEQ_ATTN, pmce_seq_attention_ex_f32(w.QKV, w.AO, B * J, T, C, J, 1, (long long)T * J, J, pk(m), stream));
  }
  const bool fuse = ln_in_product(m) && sw.proj.wp && sw.fc2.wp;
  if (fuse) {  // x += proj(attn); XN = norm2(x)
    RUN(P_GEMM_LIFTER, pmce_gemm_nt_split_f16_ln(w.AO, sw.proj.wp, 1, sw.proj.scale, bw.proj_b, w.X, (int)M, C, nullptr, nullptr, 0.f,
                                                 w.X, bw.norm2_w, bw.norm2_b, 1e-6f, w.XN, stream));
  } else {
    RUN(P_GEMM_LIFTER, lgemm(m, w.AO, bw.proj_w, sw.proj, bw.proj_b, w.X, w.X, (int)M, C,
                            C, C, C, 0, stream, pk(m)));
    RUN(P_LN, pmce_ln_chain_ex_f32(w.X, M, C, nullptr, nullptr, 0.f, nullptr, 1, 1, nullptr, bw.norm2_w,
                                   bw.norm2_b, 1e-6f, w.XN, pk(m), stream));
  }
  float* Hid = w.QKV;
  RUN(P_GEMM_LIFTER, lgemm(m, w.XN, bw.fc1_w, sw.fc1, bw.fc1_b, nullptr, Hid, (int)M,
                          2 * C, C, C, 2 * C, 1, stream, pk(m), pk(m)));
  if (fuse && post) {  // x = norm_s/t(x + fc2(h)); XN = next norm1(x)
    RUN(P_GEMM_LIFTER, pmce_gemm_nt_split_f16_ln(Hid, sw.fc2.wp, 1, sw.fc2.scale, bw.fc2_b, w.X, (int)M, 2 * C, post->w1, post->b1, 1e-6f,
                                                 w.X, post->w2, post->b2, 1e-6f, post->w2 ? w.XN : nullptr, stream));
    return PMCE_OK;
  }
  RUN(P_GEMM_LIFTER, lgemm(m, Hid, bw.fc2_w, sw.fc2, bw.fc2_b, w.X, w.X, (int)M, C,
                          2 * C, 2 * C, C, 0, stream, pk(m)));
  if (post)
    RUN(P_LN, pmce_ln_chain_ex_f32(w.X, M, C, post->w1, post->b1, 1e-6f, nullptr, J, T, w.X, post->w2, post->b2, 1e-6f,
                                   post->w2 ? w.XN : nullptr, pk(m), stream));
  return PMCE_OK;
}

// Split mode: the image features are the path's one RAW input that feeds products (imgfeat_embed, PoseEstimation.py:80; the GRU's
// layer-0 input projection, CoevoDecoder.py:228) - every other product operand is a LayerNorm / attention / GELU / GRU output.  The
// reference's Linear takes any fp32 value, so they are stored once per call as f16 planes of row * 2^-e(row) with 2^e per row
// (pmce_split_rows_scaled_f16): features of 1e5 or 1e-7 magnitude keep fp32-grade accuracy, and both products read pre-split
// operands (no split work in their k-loops).  Must run before the two branches fork.
int prep_features(pmce_model* m, const float* img_feat, int nframes, LifterWs& w, hipStream_t stream) {
  if (!m->split_now) return PMCE_OK;
  RUN(P_MISC, pmce_split_rows_scaled_f16(img_feat, nframes, F, F, w.FS, w.FR, stream));
  return PMCE_OK;
}

// a product on the row-scaled feature planes (optionally with mapped output rows)
int rs_gemm(const pmce_model* m, const float* FS, const float* FR, const SplitW& sw, const float* bias, float* Cc, int M, int N,
            long long ldc, int c_div, long long c_lo, long long c_hi, hipStream_t s) {
  return pmce_gemm_nt_split_f16_blk(FS, FR, sw.wp, sw.scale, bias, nullptr, Cc, M, N, F, F, ldc, 0, 1, 0, c_div, c_lo, c_hi, s);
}

// embedding + SpatialBlocks[0] over `nframes` frames; leaves the block output (before norm_s) in w.X
int lifter_frames(pmce_model* m, const float* pose2d, const float* img_feat, int nframes, LifterWs& w, hipStream_t stream) {
  const int J = m->J, C = m->C;
  const long long M = (long long)nframes * J;
  PMCE_REQUIRE(M < (1ll << 31), "lifter: too many tokens");
  if (m->split_now && m->s_ie.wp)
    RUN(P_GEMM_LIFTER, rs_gemm(m, w.FS, w.FR, m->s_ie, m->w.ie_b, w.E, nframes, C, C, 0, 0, 0, stream));
  else
    RUN(P_GEMM_LIFTER, lgemm(m, img_feat, m->w.ie_w, m->s_ie, m->w.ie_b, nullptr, w.E,
                            nframes, C, F, F, C, 0, stream));
  // tokens + SpatialBlocks[0].norm1 in one pass (the tokens do not travel to HBM and back between the two)
  RUN(P_EMBED, pmce_embed_ln_f32(pose2d, w.E, m->w.je_w, m->w.je_b, m->w.spos, w.X, M, J, C, m->w.blk[0][0].norm1_w,
                                 m->w.blk[0][0].norm1_b, 1e-6f, w.XN, pk(m), stream));
  return lifter_block_body(m, 0, 0, M, nframes, 0, w, stream);
}

// post-norm of block (kind, i): norm_s / norm_t (shared across depth, eps 1e-6), + temporal_pos_embed after the first
// spatial block, fused with the NEXT block's norm1
int lifter_post_norm(pmce_model* m, int kind, int i, long long M, LifterWs& w, hipStream_t stream) {
  const int J = m->J, C = m->C;
  const float* nw = kind == 0 ? m->w.ns_w : m->w.nt_w;
  const float* nb = kind == 0 ? m->w.ns_b : m->w.nt_b;
  const float* add = (kind == 0 && i == 0) ? m->w.tpos : nullptr;
  const float *w2 = nullptr, *b2 = nullptr;
  if (kind == 0) {
    w2 = m->w.blk[1][i].norm1_w;
    b2 = m->w.blk[1][i].norm1_b;
  } else if (i + 1 < m->depth) {
    w2 = m->w.blk[0][i + 1].norm1_w;
    b2 = m->w.blk[0][i + 1].norm1_b;
  }
  RUN(P_LN, pmce_ln_chain_ex_f32(w.X, M, C, nw, nb, 1e-6f, add, J, T, w.X, w2, b2, 1e-6f, w2 ? w.XN : nullptr, pk(m), stream));
  return PMCE_OK;
}

// everything from TemporalBlocks[0] on, for B clips whose tokens (after norm_s + temporal_pos_embed) are in w.X and whose
// TemporalBlocks[0].norm1 output is in w.XN
int lifter_rest(pmce_model* m, float* pose3d, int B, LifterWs& w, hipStream_t stream) {
  const int J = m->J, C = m->C;
  const long long M = (long long)B * T * J;
  bool head_pre = false;
  for (int i = 0; i < m->depth; ++i) {
    for (int kind = (i == 0 ? 1 : 0); kind < 2; ++kind) {  // (SpatialBlocks[0] - the one whose post-norm adds the position embedding - is lifter_frames')
      const PostNorm pn = post_norm_of(m, kind, i);
      // The LAST block's post-norm (norm_t, nothing after it but the head) runs inside the head when it would be a launch of its own
      // (where the product's epilogue carries it - ln_in_product - it stays there).
      const bool last = kind == 1 && i + 1 == m->depth;
      head_pre = last && !ln_in_product(m);
      PMCE_TRY(lifter_block_body(m, kind, i, M, B * T, B, w, stream, head_pre ? nullptr : &pn));
    }
  }
  RUN(P_HEAD, pmce_lifter_head_ex_f32(w.X, head_pre ? m->w.nt_w : nullptr, head_pre ? m->w.nt_b : nullptr, 1e-6f, m->w.reg0_w, m->w.reg0_b,
                                      m->w.reg1_w, m->w.reg1_b, m->w.fus_w, m->w.fus_b, pose3d, B, T, J, C, stream));
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
  const SplitW& sw = layer == 0 ? m->s_whh0 : m->s_whh1;
  const bool split = m->split_now && sw.wp;
  const float* whh = split ? sw.wp : (layer == 0 ? m->w.whh0 : m->w.whh1);  // the packed rows keep the fp32 row stride
  const float* bhh = layer == 0 ? m->w.bhh0 : m->w.bhh1;
  // (the packed rows' 2^-s: one per row, direction-major like the weight itself - a backward-only step starts at row 3 GH)
  auto step = [&](const float* gi0, const float* gi1, const float* w0, const float* w1, const float* b0, const float* b1,
                  const float* hp0, const float* hp1, float* ho0, float* ho1, int ndir) {
    const float* scale = split ? sw.scale + (w0 == whh ? 0 : 3 * GH) : nullptr;
    return split ? pmce_gru_step_split_blk_f32(gi0, gi1, w0, w1, scale, b0, b1, hp0, hp1, ho0, ho1, gi_rs, 2 * GH, B, GH, ndir, stream)
                 : pmce_gru_step_f32(gi0, gi1, w0, w1, b0, b1, hp0, hp1, ho0, ho1, gi_rs, 2 * GH, B, GH, ndir, stream);
  };
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
      RUN(P_GRU_STEP, step(gif, gib, whh, whh_b, bhh, bhh_b, hp_f, hp_b, ho_f, ho_b, 2));
    else if (af)
      RUN(P_GRU_STEP, step(gif, nullptr, whh, nullptr, bhh, nullptr, hp_f, nullptr, ho_f, nullptr, 1));
    else
      RUN(P_GRU_STEP, step(gib, nullptr, whh_b, nullptr, bhh_b, nullptr, hp_b, nullptr, ho_b, nullptr, 1));
  }
  return PMCE_OK;
}

// image-feature branch of Pose2Mesh.forward: bi-GRU + AdaLN parameters.  Depends only on img_feat, so pmce_forward
// runs it on a second stream concurrently with the pose lifter.
// recurrences + AdaLN parameters, given the layer-0 input projections GI0 (time-major [t][b][6144])
int gru_rest(pmce_model* m, int B, DecoderWs& w, hipStream_t stream);

int gru_part(pmce_model* m, const float* img_feat, const LifterWs& lw, int B, DecoderWs& w, hipStream_t stream) {
  // ---- bi-GRU over the 16 frames (CoevoDecoder.py:228); buffers are time-major [t][b][.] ----
  // layer 0 input projections for both directions in one product: rows (b,t) of img_feat -> rows (t,b) of GI0
  // (split mode: from the row-scaled planes of prep_features)
  if (m->split_now && m->s_wih0.wp)
    RUN(P_GEMM_GRU_IN, rs_gemm(m, lw.FS, lw.FR, m->s_wih0, m->w.bih0, w.GI0, B * T, 6 * GH, 6 * GH, T, (long long)B * 6 * GH, 6 * GH, stream));
  else
    RUN(P_GEMM_GRU_IN, pmce_gemm_nt_f32(img_feat, m->w.wih0, m->w.bih0, nullptr, w.GI0, B * T,
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
  SplitW wih1_b = m->s_wih1;  // rows 3072.. of the packed weight (same row stride as the fp32 one) and of its per-row 2^-s
  if (wih1_b.wp) {
    wih1_b.wp += (long long)3 * GH * 2 * GH;
    wih1_b.scale += 3 * GH;
  }
  // Split-f16 form (round 6): layer 0's output is split ONCE (34 MB at B = 256, one HBM-rate pass) instead of inside the two products' k-loops,
  // where splitting an fp32 operand costs 80 vector instructions per k-tile and wave - matrix time.  The same planes, the same results.
  const bool y0_packed = m->split_now && m->s_wih1.wp;
  const float* y0 = w.Y0;
  if (y0_packed) {
    RUN(P_GEMM_GRU_IN, pmce_split_rows_f16(w.Y0, (long long)T * B, 2 * GH, 2 * GH, w.Y0P, stream));
    y0 = w.Y0P;
  }
  RUN(P_GEMM_GRU_IN, lgemm(m, y0, m->w.wih1, m->s_wih1, m->w.bih1, nullptr, GI1f, 9 * B, 3 * GH, 2 * GH, 2 * GH,
                           3 * GH, 0, stream, y0_packed ? 1 : 0));
  RUN(P_GEMM_GRU_IN, lgemm(m, y0 + (long long)8 * B * 2 * GH, m->w.wih1 + (long long)3 * GH * 2 * GH, wih1_b,
                           m->w.bih1 + 3 * GH, nullptr, GI1b, 8 * B, 3 * GH, 2 * GH, 2 * GH, 3 * GH, 0, stream, y0_packed ? 1 : 0));
  PMCE_TRY(gru_layer(m, 1, GI1f, GI1b, 3 * GH, 0, T - 1, 9, 8, w.Y1, B, stream));
  const float* g = w.Y1 + (long long)8 * B * 2 * GH;  // img_feat = y[seqlen // 2], [B, 2048]

  // ---- all live AdaLN gamma/beta in one product (CoevoDecoder.py:19-20,27-28) ----
  RUN(P_GEMM_ADA, lgemm(m, g, m->w.ada_w, m->s_ada, m->w.ada_b, nullptr, w.GB, B, N_ADA * 128, 2 * GH, 2 * GH,
                        N_ADA * 128, 0, stream));
  return PMCE_OK;
}

// joint-side preparation of CoevoBlock k (1..3): jf, xk and the folded key/value operands of its vertex<-joint
// cross-attention.  Depends only on the joints and the AdaLN parameters, not on the vertex stream.
int joint_prep(pmce_model* m, int k, const float* joints, int B, DecoderWs& w, hipStream_t stream) {
  const int J = m->J;
  const VertexBlockW& v = m->w.vb[k - 1];
  const int ib = (k - 1) * 6, gbs = N_ADA * 128;
  // one launch: joint embedding + fold.  The f16 form of the fused vertex kernel reads the operands' image only (J <= 23); the fp32
  // form (and the two-launch fallback beyond J = 23) the fp32 operands.  jf is read by the joint stream of block 3 only.
  const bool image = pkf(m) && J <= 23;
  RUN(P_CA_FOLD, pmce_joint_prep_f32(joints, v.joint_proj_w, v.joint_proj_b, v.joint_pos, v.j2v_w, v.j2v_b, v.j2v_K,
                                     k == 3 ? w.JF[k - 1] : nullptr, w.GB, gbs, ib + 0, ib + 1, ib + 2, v.vca_wq_w, v.vca_wq_b, v.vca_wk_w,
                                     v.vca_wk_b, v.vca_wv_w, v.vca_wv_b, v.vca_proj_w, image ? nullptr : w.KF[k - 1],
                                     image ? nullptr : w.S0[k - 1], image ? nullptr : w.VF[k - 1], image ? w.CAI[k - 1] : nullptr, B, J,
                                     stream));
  return PMCE_OK;
}

// joint stream of CoevoBlock 3 (the only live one, CoevoDecoder.py:235-237): needs the block's INPUT vertices.
int joint_branch(pmce_model* m, const float* joints, const float* vt_in, float* cam_pose, int B, DecoderWs& w,
                 hipStream_t stream) {
  const int J = m->J, gbs = N_ADA * 128;
  const JointBlockW& jw = m->w.jb;
  RUN(P_TOKENS_KV, pmce_tokens_kv_pk_f32(nullptr, nullptr, vt_in, m->w.vb[2].vertx_proj_w, jw.Ev,
                                         jw.v2j_w, jw.Ek, w.GB, gbs, 19, 20,
                                         jw.jca_wk_w, jw.jca_wk_b, jw.jca_wv_w,
                                         jw.jca_wv_b, w.KVJ, B, pkf(m) ? m->tkv_img : nullptr, stream));
  const int inst[4] = {18, 21, 22, 23};
  RUN(P_JOINT_STREAM, pmce_joint_stream_f32(w.JF[2], jw.j_Q, w.KVJ, w.GB, gbs, jw.stream, inst, joints, nullptr,
                                            cam_pose, B, J, 3, stream));
  return PMCE_OK;
}

// vertex stream of CoevoBlock k (CoevoDecoder.py:183-189, vertex side): cross-attention from the joints (folded operands
// of joint_prep), its FFN, the self-attention block, coordinate head + residual.  vt_cur -> vt_next ([B,431,3]).
int vertex_block(pmce_model* m, int k, const float* vt_cur, float* vt_next, int B, DecoderWs& w, hipStream_t stream) {
  const int J = m->J, gbs = N_ADA * 128;
  const VertexBlockW& v = m->w.vb[k - 1];
  const int ib = (k - 1) * 6;  // AdaLN instances: vca.normq,normk,normv,norm2, vsa.norm1,norm2
  // CrossAttentionBlock (CoevoDecoder.py:82-87) in one launch (J > 23: the launcher's two-launch form through F1)
  RUN(P_VERTEX_CA_MLP, pmce_vertex_ca_mlp_pk_f32(nullptr, vt_cur, v.vertx_proj_w, v.Eq, w.KF[k - 1], w.S0[k - 1], w.VF[k - 1],
                                                 v.vca_proj_b, w.GB, gbs, ib + 3, v.vca_fc1_w, v.vca_fc1_b, v.vca_fc2_w,
                                                 v.vca_fc2_b, w.F2, w.F1, B, J, pkf(m), m->ffn_img[k - 1][0], w.CAI[k - 1], stream));
  if (pkf(m)) {  // AdaLN + qkv + attention + proj + residual in one launch (coevo.hip vertex_sab)
    RUN(P_VERTEX_SA, pmce_vertex_sab_split_f32(w.F2, w.GB, gbs, ib + 4, m->qkv_img[k - 1], v.vsa_qkv_b, v.vsa_proj_w, v.vsa_proj_b, w.QKV,
                                               w.F1, B, stream));
  } else {
    RUN(P_ADALN_QKV, pmce_adaln_qkv_f32(w.F2, w.GB, gbs, ib + 4, v.vsa_qkv_w, v.vsa_qkv_b, w.QKV, B, stream));
    RUN(P_VERTEX_SA, pmce_vertex_sa_f32(w.F2, w.QKV, v.vsa_proj_w, v.vsa_proj_b, w.F1, B, stream));
  }
  RUN(P_ADALN_MLP, pmce_adaln_mlp_pk_f32(w.F1, w.GB, gbs, ib + 5, v.vsa_fc1_w, v.vsa_fc1_b,
                                         v.vsa_fc2_w, v.vsa_fc2_b, nullptr,
                                         v.vcoor_w, v.vcoor_b, vt_cur, vt_next, B, pkf(m), m->ffn_img[k - 1][1], stream));
  return PMCE_OK;
}

// joint/vertex branch of Pose2Mesh.forward (needs the joints and the outputs of gru_part).  With a side stream the
// joint-side work of blocks 2-3 and the joint stream of block 3 run beside the vertex stream (they are short,
// latency-bound kernels); side == nullptr runs everything in order on `stream`.
int coevo_part(pmce_model* m, const float* joints, float* cam_pose, float* cam_mesh, int B, DecoderWs& w,
               hipStream_t stream, hipStream_t side) {
  const int J = m->J;
  const float* g = w.Y1 + (long long)8 * B * 2 * GH;
  if (side) {
    PMCE_TRY(ev_record(m->ev_a, stream, "coevo fork a"));  // joints and AdaLN parameters are ready
    PMCE_TRY(ev_wait(side, m->ev_a, "coevo fork a"));
    PMCE_TRY(joint_prep(m, 2, joints, B, w, side));
    PMCE_TRY(joint_prep(m, 3, joints, B, w, side));
    PMCE_TRY(ev_record(m->ev_b, side, "coevo join b"));
  }
  // ---- vertex init (CoevoDecoder.py:232) ----
  RUN(P_GATHER, pmce_vertex_init_gather_f32(joints, m->w.vj, w.VT[0], B, J, stream));
  PMCE_TRY(joint_prep(m, 1, joints, B, w, stream));
  float* vt_cur = w.VT[0];
  for (int k = 1; k <= 3; ++k) {
    float* vt_next = w.VT[k % 3];
    if (k > 1) {
      if (side) {
        if (k == 2) PMCE_TRY(ev_wait(stream, m->ev_b, "coevo join b"));
      } else {
        PMCE_TRY(joint_prep(m, k, joints, B, w, stream));
      }
    }
    if (k == 3) {
      if (side) {
        PMCE_TRY(ev_record(m->ev_c, stream, "coevo fork c"));  // block-3 input vertices are ready
        PMCE_TRY(ev_wait(side, m->ev_c, "coevo fork c"));
        PMCE_TRY(joint_branch(m, joints, vt_cur, cam_pose, B, w, side));
        PMCE_TRY(ev_record(m->ev_d, side, "coevo join d"));
      }
    }
    PMCE_TRY(vertex_block(m, k, vt_cur, vt_next, B, w, stream));
    if (k == 3 && !side) PMCE_TRY(joint_branch(m, joints, vt_cur, cam_pose, B, w, stream));
    vt_cur = vt_next;
  }
  // ---- 431 -> 6890 upsample conv + 3 residual Linear(2048->6890) as ONE product (CoevoDecoder.py:238-244) ----
  // (split mode: the operand is written pre-split - the product's k-loop spends no vector instruction on splitting it)
  const int fa_packed = (m->split_now && m->s_final.wp) ? 1 : 0;
  RUN(P_FINAL_OP, pmce_build_final_operand_pk_f32(g, vt_cur, w.FA, B, FINAL_K, fa_packed, stream));
  RUN(P_GEMM_FINAL, lgemm(m, w.FA, m->w.final_w, m->s_final, m->w.final_b, nullptr, cam_mesh, B, NVF * 3, FINAL_K,
                          FINAL_K, NVF * 3, 0, stream, fa_packed));
  if (side) PMCE_TRY(ev_wait(stream, m->ev_d, "coevo join d"));  // cam_pose is written by the side stream
  return PMCE_OK;
}

// second stream + fork/join events, created on first use
// pmce_model_wait_lifter: the point of a forward after which only its decoder is left
int mark_lifter_done(pmce_model* m, hipStream_t stream) {
  if (!m->ev_lifter && hipEventCreateWithFlags(&m->ev_lifter, hipEventDisableTiming) != hipSuccess) {
    m->ev_lifter = nullptr;
    pmce_set_error("could not create the lifter-done event: %s", hipGetErrorString(hipGetLastError()));
    return PMCE_ERR_LAUNCH;
  }
  return ev_record(m->ev_lifter, stream, "lifter done");
}

// A launch sequence that failed after its fork leaves the side stream un-joined: the next forward on this handle would
// race it on the shared workspace.  Drain both streams before reporting the error (error path only; never taken inside a
// successful, capturable forward).
int fail_after_fork(pmce_model* m, hipStream_t stream, int rc) {
  if (m->side) (void)hipStreamSynchronize(m->side);
  (void)hipStreamSynchronize(stream);
  return rc;
}

// A wave executing f16 matrix instructions disturbs packed-fp32 arithmetic of OTHER waves on the same CU (measured:
// scripts/microbench/victims.py).  No kernel of this library contains packed-fp32 instructions (build.py), so the
// split-f16 form keeps the two-stream schedule; split_overlap = false (PMCE_SPLIT_OVERLAP=0) sends everything down one stream.
bool two_streams(const pmce_model* m) { return m->concurrent && (!m->split_now || m->split_overlap); }

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
namespace {
// (Re)build the packed f16 planes of every large weight in model-owned memory, or drop them (fp32 mode).  A load-time step, like
// the packing the host side does: runs on `stream` - the stream the caller produced the fp32 weights on (a torch side stream is
// non-blocking: the null stream would NOT be ordered behind it) - and waits for that stream only.
struct SplitItem { const float* w; int n, k; SplitW* dst; };
size_t split_item_floats(int n, int k) { return ((((size_t)n + 63) & ~(size_t)63) * k) + (((size_t)n + 63) & ~(size_t)63); }  // planes (rows padded to the 64-row blocks of the blocked layout) + 2^-s per row
size_t ffn_img_floats() { return ((size_t)pmce_ffn_image_floats() + 63) & ~(size_t)63; }  // one FFN's LDS image (coevo.hip), 256-byte granules
size_t qkv_img_floats() { return ((size_t)pmce_qkv_image_floats() + 63) & ~(size_t)63; }
size_t tkv_img_floats() { return ((size_t)pmce_tkv_image_floats() + 63) & ~(size_t)63; }
// bytes of the planes of a model with / without lifter and decoder
size_t split_bytes_for(int C, int depth, bool lifter, bool decoder) {
  size_t f = 0;
  if (lifter) {
    f += split_item_floats(C, F);
    f += 2 * (size_t)depth * (split_item_floats(3 * C, C) + split_item_floats(C, C) + split_item_floats(2 * C, C) + split_item_floats(C, 2 * C));
  }
  if (decoder)
    f += split_item_floats(6 * GH, F) + split_item_floats(6 * GH, 2 * GH) + 2 * split_item_floats(6 * GH, GH) +
         split_item_floats(N_ADA * 128, 2 * GH) + split_item_floats(NVF * 3, FINAL_K) + 6 * ffn_img_floats() + 3 * qkv_img_floats() +
         tkv_img_floats();
  return f * sizeof(float);
}
int build_split_weights(pmce_model* m, hipStream_t stream) {
  if (m->split_adopted && m->split_gemm && m->split_arena) return PMCE_OK;  // another handle's planes of the same weights
  m->split_adopted = false;
  if (m->split_arena) {
    (void)hipDeviceSynchronize();  // forwards in flight (any stream, any lane) may still read the old planes; not capturable
    m->split_arena.reset();        // frees them unless another handle shares them
  }
  for (auto& kind : m->sblk)
    for (auto& b : kind) b = LifterBlockSplit{};
  m->s_ie = m->s_wih0 = m->s_wih1 = m->s_whh0 = m->s_whh1 = m->s_ada = m->s_final = SplitW{};
  for (auto& b : m->ffn_img) b[0] = b[1] = nullptr;
  for (auto& q : m->qkv_img) q = nullptr;
  m->tkv_img = nullptr;
  if (!m->split_gemm) return PMCE_OK;
  const int C = m->C;
  std::vector<SplitItem> items;
  if (m->has_lifter) {
    items.push_back({m->w.ie_w, C, F, &m->s_ie});
    for (int kind = 0; kind < 2; ++kind)
      for (int i = 0; i < m->depth; ++i) {
        const LifterBlockW& bw = m->w.blk[kind][i];
        LifterBlockSplit& sw = m->sblk[kind][i];
        items.push_back({bw.qkv_w, 3 * C, C, &sw.qkv});
        items.push_back({bw.proj_w, C, C, &sw.proj});
        items.push_back({bw.fc1_w, 2 * C, C, &sw.fc1});
        items.push_back({bw.fc2_w, C, 2 * C, &sw.fc2});
      }
  }
  if (m->has_decoder) {
    items.push_back({m->w.wih0, 6 * GH, F, &m->s_wih0});
    items.push_back({m->w.wih1, 6 * GH, 2 * GH, &m->s_wih1});
    items.push_back({m->w.whh0, 6 * GH, GH, &m->s_whh0});  // recurrent weights, both directions: gru_step's three-product form
    items.push_back({m->w.whh1, 6 * GH, GH, &m->s_whh1});
    items.push_back({m->w.ada_w, N_ADA * 128, 2 * GH, &m->s_ada});
    items.push_back({m->w.final_w, NVF * 3, FINAL_K, &m->s_final});
  }
  if (items.empty()) return PMCE_OK;
  size_t floats = 0;
  for (auto& it : items) floats += split_item_floats(it.n, it.k);
  if (m->has_decoder) floats += 6 * ffn_img_floats() + 3 * qkv_img_floats() + tkv_img_floats();
  if (m->caller_arena) {  // the caller's memory (its allocator, its lifetime): pmce_model_set_split_arena
    if (m->caller_arena_bytes < floats * sizeof(float)) {
      pmce_set_error("model_finalize: the split arena holds %zu bytes, the planes need %zu (pmce_model_split_bytes)", m->caller_arena_bytes,
                     floats * sizeof(float));
      return PMCE_ERR_WORKSPACE;
    }
    m->split_arena = std::shared_ptr<float>(m->caller_arena, [](float*) {});
    m->caller_arena = nullptr;
    m->caller_arena_bytes = 0;
  } else {
    float* arena = nullptr;
    const hipError_t rc = hipMalloc(reinterpret_cast<void**>(&arena), floats * sizeof(float));
    if (rc != hipSuccess) {
      pmce_set_error("model_finalize: hipMalloc(%zu bytes) for the split weights failed: %s", floats * sizeof(float), hipGetErrorString(rc));
      return PMCE_ERR_LAUNCH;
    }
    m->split_arena = std::shared_ptr<float>(arena, [](float* q) { (void)hipFree(q); });
  }
  float* p = m->split_arena.get();
  for (auto& it : items) {
    float* wp = p;
    float* sc = p + ((((size_t)it.n + 63) & ~(size_t)63) * it.k);
    p = wp + split_item_floats(it.n, it.k);
    // every weight in the blocked layout (what a tile - or a GRU step's workgroup - fetches per k-tile is contiguous)
    PMCE_TRY(pmce_gemm_pack_split_f16_blk(it.w, it.n, it.k, it.k, wp, sc, stream));
    it.dst->wp = wp;
    it.dst->scale = sc;
  }
  if (m->has_decoder)  // the decoder FFNs' and qkv weights' LDS images (coevo.hip): every workgroup of their launches copies one (LDS-DMA) instead of converting
    for (int k = 0; k < 3; ++k) {
      const VertexBlockW& v = m->w.vb[k];
      PMCE_TRY(pmce_ffn_pack_f16(v.vca_fc1_w, v.vca_fc2_w, p, stream));
      m->ffn_img[k][0] = p;
      p += ffn_img_floats();
      PMCE_TRY(pmce_ffn_pack_f16(v.vsa_fc1_w, v.vsa_fc2_w, p, stream));
      m->ffn_img[k][1] = p;
      p += ffn_img_floats();
      PMCE_TRY(pmce_qkv_pack_f16(v.vsa_qkv_w, p, stream));
      m->qkv_img[k] = p;
      p += qkv_img_floats();
    }
  if (m->has_decoder) {
    PMCE_TRY(pmce_tkv_pack_f16(m->w.jb.v2j_w, m->w.jb.jca_wk_w, m->w.jb.jca_wv_w, p, stream));
    m->tkv_img = p;
    p += tkv_img_floats();
  }
  if (hipStreamSynchronize(stream) != hipSuccess) {
    pmce_set_error("model_finalize: packing the split weights failed");
    return PMCE_ERR_LAUNCH;
  }
  return PMCE_OK;
}
}  // namespace

namespace {
// the sticky overflow word: pinned host memory the device can write (made at finalize: creating a handle needs no GPU)
int ensure_overflow_word(pmce_model* m) {
  if (m->oflow) return PMCE_OK;
  unsigned* w = nullptr;
  if (hipHostMalloc(reinterpret_cast<void**>(&w), sizeof(unsigned), hipHostMallocMapped) != hipSuccess || !w) {
    (void)hipGetLastError();
    pmce_set_error("model_finalize: hipHostMalloc of the overflow word failed");
    return PMCE_ERR_LAUNCH;
  }
  *w = 0u;
  m->oflow = std::shared_ptr<unsigned>(w, [](unsigned* q) { (void)hipHostFree(q); });
  return PMCE_OK;
}
}  // namespace

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
  m->split_gemm = pmce_env_int("PMCE_SPLIT_F16", 1) != 0;
  m->split_min_batch = pmce_env_int("PMCE_SPLIT_MIN_BATCH", 1);
  m->split_overlap = pmce_env_int("PMCE_SPLIT_OVERLAP", 1) != 0;
  m->strict_overflow = pmce_env_int("PMCE_STRICT_OVERFLOW", 0) != 0;
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
  m->split_adopted = false;
  return PMCE_OK;
}

int pmce_model_set_regressor_rows(pmce_model* m, int rows) {
  PMCE_REQUIRE(m && rows > 0 && rows <= 32, "model_set_regressor_rows: rows must be in 1..32");
  m->jr_rows = rows;
  return PMCE_OK;
}

size_t pmce_model_split_bytes(const pmce_model* m) {
  if (!m || (m->split_adopted && m->split_arena)) return 0;
  bool lifter = false, decoder = false;
  for (auto& s : m->names)
    if (m->ptr.count(s)) (s.rfind("lifter.", 0) == 0 ? lifter : decoder) = true;
  return split_bytes_for(m->C, m->depth, lifter, decoder);
}
int pmce_model_set_split_arena(pmce_model* m, void* arena, size_t bytes) {
  PMCE_REQUIRE(m, "model_set_split_arena: null model");
  PMCE_REQUIRE(arena == nullptr || (reinterpret_cast<uintptr_t>(arena) & 255) == 0, "model_set_split_arena: the arena must be 256-byte aligned");
  m->caller_arena = static_cast<float*>(arena);
  m->caller_arena_bytes = arena ? bytes : 0;
  return PMCE_OK;
}
int pmce_model_finalize(pmce_model* m) { return pmce_model_finalize_on(m, nullptr); }
int pmce_model_finalize_on(pmce_model* m, pmce_stream_t stream) {
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
  // resolve every registered name into its field of m->w: launch sequences read plain pointers from here on
  for (auto& sl : m->slots) {
    auto it = m->ptr.find(sl.name);
    *sl.dst = it == m->ptr.end() ? nullptr : it->second;
  }
  PMCE_TRY(ensure_overflow_word(m));
  PMCE_TRY(build_split_weights(m, stream));
  m->finalized = true;
  return PMCE_OK;
}

int pmce_model_set_gemm_mode(pmce_model* m, int split_f16) { return pmce_model_set_gemm_mode_on(m, split_f16, nullptr); }
int pmce_model_set_gemm_mode_on(pmce_model* m, int split_f16, pmce_stream_t stream) {
  PMCE_REQUIRE(m, "model_set_gemm_mode: null model");
  if (m->split_gemm != (split_f16 != 0)) {
    m->split_gemm = split_f16 != 0;
    if (m->finalized) PMCE_TRY(build_split_weights(m, stream));
  }
  return PMCE_OK;
}
int pmce_model_gemm_mode(const pmce_model* m) { return m && m->split_gemm ? 1 : 0; }
int pmce_model_share_split_weights(pmce_model* dst, const pmce_model* src) {
  PMCE_REQUIRE(dst && src && src->finalized, "model_share_split_weights: need a destination and a finalized source");
  PMCE_REQUIRE(dst->J == src->J && dst->C == src->C && dst->depth == src->depth, "model_share_split_weights: different configurations");
  for (auto& n : dst->names) {  // the same weights, tensor by tensor
    const auto a = static_cast<const pmce_model*>(dst)->ptr.find(n);
    const auto b = src->ptr.find(n);
    PMCE_REQUIRE((a == dst->ptr.end()) == (b == src->ptr.end()) && (a == dst->ptr.end() || a->second == b->second),
                 "model_share_split_weights: tensor '%s' differs between the two handles", n.c_str());
  }
  if (!src->split_arena) return PMCE_OK;  // fp32 mode: nothing to share
  dst->split_arena = src->split_arena;
  for (int k = 0; k < 2; ++k)
    for (int i = 0; i < 8; ++i) dst->sblk[k][i] = src->sblk[k][i];
  dst->s_ie = src->s_ie; dst->s_wih0 = src->s_wih0; dst->s_wih1 = src->s_wih1; dst->s_whh0 = src->s_whh0; dst->s_whh1 = src->s_whh1;
  dst->s_ada = src->s_ada; dst->s_final = src->s_final;
  for (int k = 0; k < 3; ++k) dst->ffn_img[k][0] = src->ffn_img[k][0], dst->ffn_img[k][1] = src->ffn_img[k][1], dst->qkv_img[k] = src->qkv_img[k];
  dst->tkv_img = src->tkv_img;
  dst->split_gemm = true;
  dst->split_adopted = true;
  dst->oflow = src->oflow;  // lanes of one model report to one word
  dst->strict_overflow = src->strict_overflow;  // ... and follow its policy and its small-batch threshold
  dst->split_min_batch = src->split_min_batch;
  return PMCE_OK;
}
int pmce_model_overflowed(const pmce_model* m) {
  return m && m->oflow && *reinterpret_cast<const volatile unsigned*>(m->oflow.get()) != 0u ? 1 : 0;
}
int pmce_model_clear_overflow(pmce_model* m) {
  PMCE_REQUIRE(m && m->oflow, "model_clear_overflow: null model");
  *reinterpret_cast<volatile unsigned*>(m->oflow.get()) = 0u;
  return PMCE_OK;
}
int pmce_model_set_clock_probe(pmce_model* m, unsigned long long* device_two_words) {
  PMCE_REQUIRE(m, "model_set_clock_probe: null model");
  m->clk = device_two_words;
  return PMCE_OK;
}
int pmce_model_set_overflow_policy(pmce_model* m, int strict) {
  PMCE_REQUIRE(m, "model_set_overflow_policy: null model");
  m->strict_overflow = strict != 0;
  return PMCE_OK;
}
int pmce_model_set_split_min_batch(pmce_model* m, int clips) {
  PMCE_REQUIRE(m && clips >= 1, "model_set_split_min_batch: need a model and clips >= 1");
  m->split_min_batch = clips;
  return PMCE_OK;
}
// the values in force (their defaults come from the environment at create time: a caller that changes one temporarily restores THIS)
int pmce_model_get_split_min_batch(const pmce_model* m) { return m ? m->split_min_batch : -1; }
int pmce_model_get_concurrency(const pmce_model* m) { return m ? (m->concurrent ? 1 : 0) : -1; }
int pmce_model_get_split_overlap(const pmce_model* m) { return m ? (m->split_overlap ? 1 : 0) : -1; }
int pmce_model_get_overflow_policy(const pmce_model* m) { return m ? (m->strict_overflow ? 1 : 0) : -1; }

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
  else if (n == "Y1") p = dw.Y1;                                           // GRU layer-1 output [16,B,2048] (pruned steps unset)
  else if (n == "g") p = dw.Y1 + (long long)8 * batch * 2 * GH;            // y[8] [B,2048]
  else if (n == "GB") p = dw.GB;                                           // AdaLN gamma|beta [B,24*128]
  else if (n == "VT0") p = dw.VT[0];                                       // after a forward: v3
  else if (n == "VT1") p = dw.VT[1];                                       // v1
  else if (n == "VT2") p = dw.VT[2];                                       // v2
  else if (n == "F1") p = dw.F1;
  else if (n == "F2") p = dw.F2;
  else if (n == "JM") p = dw.JM;
  else return -1;
  return (long long)reinterpret_cast<uintptr_t>(p);  // the carver ran on a null base: the pointer value IS the offset
}

namespace {
struct SinkGuard {  // the launchers of this thread report to the model's flag (and clock probe) only while one of its entry points runs
  ~SinkGuard() {
    pmce_set_overflow_sink(nullptr);
    pmce_set_clock_sink(nullptr);
  }
};
}  // namespace

static int check_ws(pmce_model* m, int batch, void* ws, size_t ws_bytes) {
  PMCE_REQUIRE(m && m->finalized, "model not finalized (call pmce_model_finalize after registering all tensors)");
  if (m->strict_overflow && m->oflow && *reinterpret_cast<volatile unsigned*>(m->oflow.get()) != 0u) {
    pmce_set_error("strict overflow policy: an earlier call on this model produced non-finite values in a product of the split-f16 form "
                   "(non-finite inputs, an intermediate activation beyond the f16 range |a| > 65504, or fp32 overflow).  "
                   "pmce_model_clear_overflow re-arms the model; pmce_model_set_gemm_mode(m, 0) computes on the fp32 matrix pipe, which "
                   "has fp32's range");
    return PMCE_ERR_OVERFLOW;
  }
  PMCE_REQUIRE(batch > 0, "batch must be positive");
  PMCE_REQUIRE(ws && (reinterpret_cast<uintptr_t>(ws) & 255) == 0, "workspace must be non-null and 256-byte aligned");
  if (ws_bytes < pmce_model_workspace_bytes(m, batch)) {
    pmce_set_error("workspace too small: %zu < %zu bytes for batch %d", ws_bytes, pmce_model_workspace_bytes(m, batch), batch);
    return PMCE_ERR_WORKSPACE;
  }
  m->split_now = m->split_gemm && batch >= m->split_min_batch;  // arithmetic (and with it the stream schedule) of this call
  pmce_set_overflow_sink(m->oflow.get());  // (thread-local; the entry point clears it again through its SinkGuard)
  pmce_set_clock_sink(m->clk);
  return PMCE_OK;
}

int pmce_lifter_forward(pmce_model* m, const float* pose2d, const float* img_feat, float* pose3d, int batch, void* ws,
                        size_t ws_bytes, pmce_stream_t stream) {
  PMCE_TRY(check_ws(m, batch, ws, ws_bytes));
  SinkGuard sink_guard;
  PMCE_REQUIRE(m->has_lifter, "lifter_forward: lifter tensors not registered");
  PMCE_REQUIRE(pose2d && img_feat && pose3d, "lifter_forward: null pointer");
  Carver c(ws, ws_bytes);
  LifterWs lw;
  carve_lifter(c, m, batch, lw);
  PMCE_TRY(prep_features(m, img_feat, batch * T, lw, stream));
  return lifter_impl(m, pose2d, img_feat, pose3d, batch, lw, stream);
}

int pmce_decoder_forward(pmce_model* m, const float* joints, const float* img_feat, float* cam_pose, float* cam_mesh,
                         int batch, void* ws, size_t ws_bytes, pmce_stream_t stream) {
  PMCE_TRY(check_ws(m, batch, ws, ws_bytes));
  SinkGuard sink_guard;
  PMCE_REQUIRE(m->has_decoder, "decoder_forward: decoder tensors not registered");
  PMCE_REQUIRE(joints && img_feat && cam_pose && cam_mesh, "decoder_forward: null pointer");
  Carver c(ws, ws_bytes);
  LifterWs lw;
  DecoderWs dw;
  carve_lifter(c, m, batch, lw);
  carve_decoder(c, m, batch, dw);
  PMCE_TRY(prep_features(m, img_feat, batch * T, lw, stream));
  PMCE_TRY(gru_part(m, img_feat, lw, batch, dw, stream));
  if (two_streams(m)) PMCE_TRY(ensure_side(m));
  const int rc = coevo_part(m, joints, cam_pose, cam_mesh, batch, dw, stream, two_streams(m) ? m->side : nullptr);
  return rc == PMCE_OK ? rc : fail_after_fork(m, stream, rc);
}

// CoevoBlock.forward for ONE block k in 1..3 (CoevoDecoder.py:175-191) on explicit inputs: joints[B,J,3] (m), vt_in[B,431,3],
// g[B,2048] (the AdaLN conditioning, y[seqlen//2]) -> vt_out[B,431,3] and, for k == 3 (the only block whose joint stream is
// live at inference), joint_out[B,J,3].  Operator-level entry for parity tests against the reference module's own outputs.
int pmce_coevo_block_forward(pmce_model* m, int k, const float* joints, const float* vt_in, const float* g, float* vt_out,
                             float* joint_out, int batch, void* ws, size_t ws_bytes, pmce_stream_t stream) {
  PMCE_TRY(check_ws(m, batch, ws, ws_bytes));
  SinkGuard sink_guard;
  PMCE_REQUIRE(m->has_decoder, "coevo_block_forward: decoder tensors not registered");
  PMCE_REQUIRE(k >= 1 && k <= 3, "coevo_block_forward: k must be 1, 2 or 3");
  PMCE_REQUIRE(joints && vt_in && g && vt_out, "coevo_block_forward: null pointer");
  PMCE_REQUIRE(!joint_out || k == 3, "coevo_block_forward: the joint stream is live in block 3 only (CoevoDecoder.py:235-237)");
  Carver c(ws, ws_bytes);
  LifterWs lw;
  DecoderWs dw;
  carve_lifter(c, m, batch, lw);
  carve_decoder(c, m, batch, dw);
  RUN(P_GEMM_ADA, lgemm(m, g, m->w.ada_w, m->s_ada, m->w.ada_b, nullptr, dw.GB, batch, N_ADA * 128, 2 * GH, 2 * GH, N_ADA * 128, 0, stream));
  PMCE_TRY(joint_prep(m, k, joints, batch, dw, stream));
  if (joint_out) PMCE_TRY(joint_branch(m, joints, vt_in, joint_out, batch, dw, stream));
  return vertex_block(m, k, vt_in, vt_out, batch, dw, stream);
}

static int forward_impl(pmce_model* m, const float* pose2d, const float* img_feat, float* cam_mesh, float* cam_pose,
                        float* pose3d, float* pred_pose, int batch, LifterWs& lw, DecoderWs& dw, hipStream_t stream) {
  // Fork: the GRU / AdaLN-parameter branch depends only on img_feat; it runs on a second (high-priority) stream
  // under the pose lifter, whose long matrix-core kernels leave the gaps its 25 short dependent steps need.
  // pmce_model_set_concurrency(m, 0) (or PMCE_SINGLE_STREAM=1 at create time) keeps everything on one stream.
  const bool single = !two_streams(m);
  PMCE_TRY(prep_features(m, img_feat, batch * T, lw, stream));  // read by both branches
  if (!single) {
    PMCE_TRY(ev_record(m->ev_fork, stream, "forward fork"));
    PMCE_TRY(ev_wait(m->side, m->ev_fork, "forward fork"));
    PMCE_TRY(gru_part(m, img_feat, lw, batch, dw, m->side));
    PMCE_TRY(ev_record(m->ev_join, m->side, "forward join"));
  } else {
    PMCE_TRY(gru_part(m, img_feat, lw, batch, dw, stream));
  }
  PMCE_TRY(lifter_impl(m, pose2d, img_feat, pose3d, batch, lw, stream));
  // pose3d.reshape(-1, J, 3) / 1000  (PMCE.py:17-18)
  RUN(P_MISC, pmce_div_scalar_f32(pose3d, dw.JM, (long long)batch * m->J * 3, 1000.0f, stream));
  PMCE_TRY(mark_lifter_done(m, stream));
  if (!single) PMCE_TRY(ev_wait(stream, m->ev_join, "forward join"));
  PMCE_TRY(coevo_part(m, dw.JM, cam_pose, cam_mesh, batch, dw, stream, single ? nullptr : m->side));
  if (pred_pose)
    RUN(P_JREG, pmce_j_regress_f32(cam_mesh, m->jr_indptr, m->jr_indices, m->jr_data, pred_pose, batch, m->jr_rows, NVF,
                                   1000.0f, stream));
  return PMCE_OK;
}

int pmce_forward(pmce_model* m, const float* pose2d, const float* img_feat, float* cam_mesh, float* cam_pose,
                 float* pose3d, float* pred_pose, int batch, void* ws, size_t ws_bytes, pmce_stream_t stream) {
  PMCE_TRY(check_ws(m, batch, ws, ws_bytes));
  SinkGuard sink_guard;
  PMCE_REQUIRE(m->has_lifter && m->has_decoder, "forward: needs both lifter and decoder tensors");
  PMCE_REQUIRE(pose2d && img_feat && cam_mesh && cam_pose && pose3d, "forward: null pointer");
  PMCE_REQUIRE(!pred_pose || (m->jr_indptr && m->jr_indices && m->jr_data && m->jr_rows > 0),
               "forward: pred_pose requested but no J_regressor registered (jreg.indptr/indices/data + rows)");
  Carver c(ws, ws_bytes);
  LifterWs lw;
  DecoderWs dw;
  carve_lifter(c, m, batch, lw);
  carve_decoder(c, m, batch, dw);
  if (two_streams(m)) PMCE_TRY(ensure_side(m));
  const int rc = forward_impl(m, pose2d, img_feat, cam_mesh, cam_pose, pose3d, pred_pose, batch, lw, dw, stream);
  return rc == PMCE_OK ? rc : fail_after_fork(m, stream, rc);
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
  SinkGuard sink_guard;
  PMCE_REQUIRE(m->has_lifter && m->has_decoder, "stream_precompute: needs both lifter and decoder tensors");
  PMCE_REQUIRE(pose2d_frames && feat_frames && x0 && gi0, "stream_precompute: null pointer");
  Carver c(ws, ws_bytes);
  LifterWs lw;
  carve_lifter(c, m, bf, lw);
  // window-independent lifter work: embedding + SpatialBlocks[0] + norm_s, once per frame (PoseEstimation.py:78-85)
  PMCE_TRY(prep_features(m, feat_frames, L, lw, stream));
  PMCE_TRY(lifter_frames(m, pose2d_frames, feat_frames, L, lw, stream));
  RUN(P_LN, pmce_ln_chain_f32(lw.X, (long long)L * m->J, m->C, m->w.ns_w, m->w.ns_b, 1e-6f,
                              nullptr, 1, 1, x0, nullptr, nullptr, 0.f, nullptr, stream));
  // window-independent GRU work: layer-0 input projections of both directions, once per frame (CoevoDecoder.py:216-221)
  if (m->split_now && m->s_wih0.wp)
    RUN(P_GEMM_GRU_IN, rs_gemm(m, lw.FS, lw.FR, m->s_wih0, m->w.bih0, gi0, L, 6 * GH, 6 * GH, 0, 0, 0, stream));
  else
    RUN(P_GEMM_GRU_IN, lgemm(m, feat_frames, m->w.wih0, m->s_wih0, m->w.bih0, nullptr, gi0, L, 6 * GH, F, F, 6 * GH,
                             0, stream));
  return PMCE_OK;
}

static int stream_forward_impl(pmce_model* m, const float* x0, const float* gi0, const int* win, int W, int L,
                               float* cam_mesh, float* cam_pose, float* pose3d, float* pred_pose, LifterWs& lw, DecoderWs& dw,
                               hipStream_t stream) {
  const bool single = !two_streams(m);
  hipStream_t gs = single ? stream : m->side;
  if (!single) {
    PMCE_TRY(ev_record(m->ev_fork, stream, "stream_forward fork"));
    PMCE_TRY(ev_wait(m->side, m->ev_fork, "stream_forward fork"));
  }
  {
    hipStream_t stream_save = stream;
    stream = gs;  // RUN() launches on `stream`
    RUN(P_MISC, pmce_window_rows_f32(gi0, win, dw.GI0, W, L, T, 6 * GH, stream));
    stream = stream_save;
  }
  PMCE_TRY(gru_rest(m, W, dw, gs));
  if (!single) PMCE_TRY(ev_record(m->ev_join, m->side, "stream_forward join"));
  RUN(P_LN, pmce_window_tokens_ex_f32(x0, win, m->w.tpos, m->w.blk[1][0].norm1_w, m->w.blk[1][0].norm1_b, 1e-6f, lw.X, lw.XN, W, L,
                                   T, m->J, m->C, pk(m), stream));
  PMCE_TRY(lifter_rest(m, pose3d, W, lw, stream));
  RUN(P_MISC, pmce_div_scalar_f32(pose3d, dw.JM, (long long)W * m->J * 3, 1000.0f, stream));
  PMCE_TRY(mark_lifter_done(m, stream));
  if (!single) PMCE_TRY(ev_wait(stream, m->ev_join, "stream_forward join"));
  PMCE_TRY(coevo_part(m, dw.JM, cam_pose, cam_mesh, W, dw, stream, single ? nullptr : m->side));
  if (pred_pose)
    RUN(P_JREG, pmce_j_regress_f32(cam_mesh, m->jr_indptr, m->jr_indices, m->jr_data, pred_pose, W, m->jr_rows, NVF, 1000.0f,
                                   stream));
  return PMCE_OK;
}

int pmce_stream_forward(pmce_model* m, const float* x0, const float* gi0, const int* win, int W, int L, float* cam_mesh,
                        float* cam_pose, float* pose3d, float* pred_pose, void* ws, size_t ws_bytes, pmce_stream_t stream) {
  PMCE_TRY(check_ws(m, W, ws, ws_bytes));
  SinkGuard sink_guard;
  PMCE_REQUIRE(m->has_lifter && m->has_decoder, "stream_forward: needs both lifter and decoder tensors");
  PMCE_REQUIRE(x0 && gi0 && win && cam_mesh && cam_pose && pose3d && L > 0, "stream_forward: null pointer");
  PMCE_REQUIRE(!pred_pose || (m->jr_indptr && m->jr_indices && m->jr_data && m->jr_rows > 0),
               "stream_forward: no J_regressor registered");
  Carver c(ws, ws_bytes);
  LifterWs lw;
  DecoderWs dw;
  carve_lifter(c, m, W, lw);
  carve_decoder(c, m, W, dw);
  if (two_streams(m)) PMCE_TRY(ensure_side(m));
  const int rc = stream_forward_impl(m, x0, gi0, win, W, L, cam_mesh, cam_pose, pose3d, pred_pose, lw, dw, stream);
  return rc == PMCE_OK ? rc : fail_after_fork(m, stream, rc);
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
