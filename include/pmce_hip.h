/* libpmce_hip.so — C ABI of the MI355X-native PMCE per-clip inference hot path.
 *
 * The reference (kasvii/PMCE) is pure Python/PyTorch and has no FFI; its boundary for this path is the
 * nn.Module API of lib/models (PMCE.py:15-26, PoseEstimation.py:95-120, CoevoDecoder.py:226-252) plus the
 * caller's J_regressor projection (lib/core/base.py:223-225).  The entry points below are what a binding of
 * that path needs; pmce_amd/_lib.py is the ctypes binding, pmce_amd/models/ the nn.Module-shaped host side,
 * INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions (all functions):
 *   - return 0 on success, a negative PMCE_ERR_* otherwise; never throw, never exit; the message of the last
 *     error of the calling thread is pmce_last_error_string();
 *   - every pointer is a DEVICE pointer to contiguous row-major fp32 (int32 where noted), 16-byte aligned;
 *   - the caller owns every buffer: weights, activations, the workspace and - split_f16 mode - the arena that holds the f16
 *     planes of the large weights (pmce_model_split_bytes / pmce_model_set_split_arena).  What a model handle allocates for
 *     itself: 4 bytes of pinned host memory (the overflow word, pmce_model_overflowed), one internal HIP stream and a handful of
 *     events (two-stream execution inside a forward); and, ONLY when no arena was handed over, the planes by hipMalloc;
 *   - launches are asynchronous on the hipStream_t passed in.  The calls that synchronise: pmce_model_finalize[_on] and
 *     pmce_model_set_gemm_mode[_on] (they wait for their own packing kernels on the stream given; a mode change that drops
 *     existing planes first waits for the whole device - forwards in flight may read them - and is therefore not capturable),
 *     pmce_model_profile_read (waits for the recorded events);
 *   - mutable state outside the handles: the thread-local error string and, while a model entry point runs, the thread-local
 *     pointer to that model's overflow word - one process per GPU or several host threads with their own streams are both fine -
 *     and the process-wide TUNING AIDS pmce_gemm_set_tuning, pmce_gemm_split_set_tuning (relaxed atomics
 *     read at launch time: meant for benchmarks and tests, not to be flipped while forwards are being enqueued elsewhere).
 * Fixed structural constants of the path: T = 16 frames, F = 2048 image-feature channels, V = 431 coarse
 * vertices, 6890 mesh vertices, D = 64 decoder channels, GRU hidden 1024, 8 lifter heads.  J <= 32,
 * C in {256, 512}.
 */
#ifndef PMCE_HIP_H
#define PMCE_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* pmce_stream_t; /* == hipStream_t */

#define PMCE_OK 0
#define PMCE_ERR_ARG (-1)
#define PMCE_ERR_LAUNCH (-2)
#define PMCE_ERR_WORKSPACE (-3)
#define PMCE_ERR_OVERFLOW (-4) /* strict overflow policy only: an earlier call produced non-finite values (pmce_model_overflowed) */

int pmce_version(void);
/* identifies the sources the library was built from (16 hex digits; pmce_amd.build.source_id()): profiler summaries record it */
const char* pmce_build_id(void);
const char* pmce_last_error_string(void);

/* ---------------------------------------------------------------------------------------------------------
 * Whole-path model object (host-side descriptor only: it stores the pointers the caller registers).
 * Replaces: models.PMCE.get_model / PMCE.forward (PMCE.py:7-26), models.PoseEstimation.get_model /
 * GraphormerNet.forward (PoseEstimation.py:95-120), models.CoevoDecoder.get_model / Pose2Mesh.forward
 * (CoevoDecoder.py:226-252), and `J_regressor[None] @ (pred_mesh*1000)` (lib/core/base.py:223-225).
 * ------------------------------------------------------------------------------------------------------- */
typedef struct pmce_model pmce_model;

/* num_joint: J (17, or 19 for COCO-input checkpoints); embed_dim: C (256|512); depth: lifter depth (3).
 * ENVIRONMENT: the library reads exactly five variables, all here, once per handle; each only sets the initial value of a setting that has an
 * API setter and a getter (tests/test_host_logic.py flips every one):
 *   PMCE_SPLIT_F16=0        initial gemm mode fp32 pipe           (pmce_model_set_gemm_mode / pmce_model_gemm_mode)
 *   PMCE_SPLIT_MIN_BATCH=n  calls below n clips stay on fp32 pipe (pmce_model_set_split_min_batch / _get_)
 *   PMCE_STRICT_OVERFLOW=1  strict overflow policy                (pmce_model_set_overflow_policy / _get_)
 *   PMCE_SINGLE_STREAM=1    no second stream inside a forward     (pmce_model_set_concurrency / _get_)
 *   PMCE_SPLIT_OVERLAP=0    diagnostic: the split mode strictly serial, one stream (pmce_model_get_split_overlap; Python's Pipeline reads it too)
 * (Rounds 2-4 carried 16 more A/B switches that kept superseded kernels reachable; round 5 removed them with those kernels.) */
int pmce_model_create(int num_joint, int embed_dim, int depth, pmce_model** out);
void pmce_model_destroy(pmce_model* m);
/* Register one packed tensor by name (names: pmce_model_tensor_name).  The pointer must stay valid. */
int pmce_model_set_tensor(pmce_model* m, const char* name, const void* dev_ptr);
/* Enumerate the packed-tensor names the model needs (i in [0, pmce_model_tensor_count)). */
int pmce_model_tensor_count(const pmce_model* m);
const char* pmce_model_tensor_name(const pmce_model* m, int i);
/* Optional caller-side projection (lib/core/base.py:196,225): register "jreg.indptr" (int32[R+1]), "jreg.indices"
 * (int32[nnz]), "jreg.data" (fp32[nnz]) with pmce_model_set_tensor and the row count R here. */
int pmce_model_set_regressor_rows(pmce_model* m, int rows);
/* Check that every tensor is registered and (split_f16 mode) pack the large weights as f16 planes (into the arena below, or memory of its own without one).
 * The packing kernels run on `stream` - pass the stream the registered tensors were produced on (they are read here) - and the
 * call returns when they are done (it synchronises that stream, nothing else).  pmce_model_finalize = ..._on(m, NULL): the null
 * stream, which is NOT ordered behind non-blocking streams (PyTorch's side streams). */
int pmce_model_finalize(pmce_model* m);
int pmce_model_finalize_on(pmce_model* m, pmce_stream_t stream);
/* The f16 planes (as many bytes again as the fp32 weights they mirror: 0.6 GB at C = 512) live in memory the CALLER provides:
 * pmce_model_split_bytes (after the last pmce_model_set_tensor; whatever the current mode; 0 for a handle that adopted another
 * one's planes) says how much, pmce_model_set_split_arena hands it over (256-byte aligned device memory that must outlive the model
 * and every handle sharing its planes) - before pmce_model_finalize[_on], and again before a pmce_model_set_gemm_mode[_on] that
 * switches a finalized model to split_f16.  Without an arena the library falls back to hipMalloc / hipFree of its own. */
size_t pmce_model_split_bytes(const pmce_model* m);
int pmce_model_set_split_arena(pmce_model* m, void* arena, size_t bytes);
/* Arithmetic of the path's 30 large products per forward (the pose lifter's Linear layers, the GRU input projections, the
 * packed AdaLN and final products): split_f16 != 0 (the default; env PMCE_SPLIT_F16=0 at create for the other) = the three-product f16 form of pmce_gemm_nt_split_f16 on weights the
 * model packs for itself at finalize, 0 = the fp32 matrix pipe.  Both meet fp32 accuracy (tests/test_gpu_ops.py measures
 * each against an fp64 product); may be called at any time between forwards.  In split_f16 mode the GRU recurrence, the decoder's
 * FFNs and 431x431 self-attention and the lifter's attention (pmce_seq_attention_split_f16, for J = 17 / 19) use the same
 * three-product form. */
int pmce_model_set_gemm_mode(pmce_model* m, int split_f16);
/* The same with the packing (when the mode changes on a finalized model) on `stream`; it first waits for the whole device
 * (forwards in flight may read the planes it frees), so it must not be called during a stream capture. */
int pmce_model_set_gemm_mode_on(pmce_model* m, int split_f16, pmce_stream_t stream);
int pmce_model_gemm_mode(const pmce_model* m);
/* A second handle on the SAME registered weights (a pipeline lane) takes the source's packed planes instead of packing its own copy:
 * call after the last pmce_model_set_tensor of `dst` and before its pmce_model_finalize; the planes live until the last handle goes. */
int pmce_model_share_split_weights(pmce_model* dst, const pmce_model* src);
/* Calls with fewer clips (windows) than this stay on the fp32 pipe even in split_f16 mode (default 1 = none do: the f16 form is
 * faster at every batch size; env PMCE_SPLIT_MIN_BATCH at create). */
int pmce_model_set_split_min_batch(pmce_model* m, int clips);
/* Range of the split_f16 form, and the word that reports leaving it.  The path's one RAW input that feeds products - img_feat -
 * is stored per call as row-scaled f16 planes (pmce_split_rows_scaled_f16), so ANY finite fp32 input is computed with fp32-grade
 * accuracy, exactly as the reference's Linear layers accept it (PoseEstimation.py:80, CoevoDecoder.py:228); every other product
 * operand is the output of a LayerNorm / attention / GELU / GRU gate / AdaLN, whose magnitude the weights bound.  What remains:
 * non-finite INPUTS (they propagate into the outputs of their own clip only, as in the reference) and intermediate activations
 * beyond f16's 65504 under exotic weights - these turn into inf / nan, never into a wrong finite value, and every non-finite value
 * of a forward reaches a product epilogue, which sets a word in pinned host memory the model owns (readable without
 * synchronisation; sticky until pmce_model_clear_overflow).  pmce_model_overflowed reflects the device work that has completed:
 * synchronise the stream first for a definite answer about a given call.  DEFAULT POLICY: the word only reports - calls are never
 * refused, the model behaves like the reference (a bad clip yields nan, the next call is unaffected); pmce_amd's Pipeline /
 * evaluator poll it when they drain and can re-run the offending batch on the fp32 pipe (pmce_model_set_gemm_mode(m, 0) /
 * pmce_model_set_split_min_batch), which has fp32's own range.  STRICT policy (pmce_model_set_overflow_policy(m, 1), env
 * PMCE_STRICT_OVERFLOW=1 at create): while the word is set every entry point returns PMCE_ERR_OVERFLOW before launching
 * anything.  Pipeline lanes created with pmce_model_share_split_weights share the source's word. */
int pmce_model_set_overflow_policy(pmce_model* m, int strict);
/* the values in force (defaults come from PMCE_STRICT_OVERFLOW / PMCE_SPLIT_MIN_BATCH at create; pmce_model_share_split_weights copies both to the lane) */
int pmce_model_get_overflow_policy(const pmce_model* m);
int pmce_model_get_split_min_batch(const pmce_model* m);
/* Measurement aid (bench.py), per MODEL: while set (null = off), every launch of the split GEMM made by an entry point of this model
 * adds, per workgroup, the shader clocks and the 100 MHz wall ticks its first wave was resident to device_two_words[0] / [1]: their
 * ratio x 0.1 is the shader clock in GHz the chip sustained under the kernel (MI355X is power-limited there: 1.6 - 1.8 GHz, not the
 * 2.4 GHz of the peak figures). */
int pmce_model_set_clock_probe(pmce_model* m, unsigned long long* device_two_words);
int pmce_model_overflowed(const pmce_model* m);
int pmce_model_clear_overflow(pmce_model* m);
/* Bytes of caller-provided workspace needed for a batch of B clips. */
size_t pmce_model_workspace_bytes(const pmce_model* m, int batch);

/* Byte offset inside the workspace of a named intermediate ("X","Y0","g","GB","VT0","VT1","VT2","F1","F2","JM"),
 * -1 if unknown — for parity tests of intermediates (after a forward VT1/VT2/VT0 hold the vertices after
 * coevoblock1/2/3, "g" the GRU feature y[8]). */
long long pmce_model_workspace_offset(const pmce_model* m, int batch, const char* name);

/* GraphormerNet.forward: pose2d[B,16,J,2], img_feat[B,16,2048] -> pose3d[B,J,3] (mm). */
int pmce_lifter_forward(pmce_model* m, const float* pose2d, const float* img_feat, float* pose3d, int batch, void* ws,
                        size_t ws_bytes, pmce_stream_t stream);
/* Pose2Mesh.forward: joints[B,J,3] (m), img_feat[B,16,2048] -> cam_pose[B,J,3], cam_mesh[B,6890,3] (m). */
int pmce_decoder_forward(pmce_model* m, const float* joints, const float* img_feat, float* cam_pose, float* cam_mesh,
                         int batch, void* ws, size_t ws_bytes, pmce_stream_t stream);
/* PMCE.forward: -> cam_mesh[B,6890,3] (m), cam_pose[B,J,3] (m), pose3d[B,J,3] (mm);
 * if pred_pose != NULL also the caller's projection pred_pose[B,R,3] = J_regressor @ (cam_mesh*1000) (mm). */
int pmce_forward(pmce_model* m, const float* pose2d, const float* img_feat, float* cam_mesh, float* cam_pose,
                 float* pose3d, float* pred_pose, int batch, void* ws, size_t ws_bytes, pmce_stream_t stream);

/* CoevoBlock.forward for ONE block k in 1..3 on explicit inputs (reference lib/models/CoevoDecoder.py:175-191):
 * joints[B,J,3] (m), vt_in[B,431,3], g[B,2048] (AdaLN conditioning = y[seqlen//2], :229) -> vt_out[B,431,3]; joint_out[B,J,3]
 * only for k == 3 (joints1/joints2 are discarded by Pose2Mesh.forward, :235-237), else NULL. */
int pmce_coevo_block_forward(pmce_model* m, int k, const float* joints, const float* vt_in, const float* g, float* vt_out,
                             float* joint_out, int batch, void* ws, size_t ws_bytes, pmce_stream_t stream);


/* Streaming (stride-1 windows of ONE long sequence, lib/_img_utils.py:27-57): the window-independent per-frame work
 * (embedding + SpatialBlocks[0] + norm_s of the lifter, PoseEstimation.py:78-85; GRU layer-0 input projections) is done
 * once per frame into x0[L,J,C] and gi0[L,6144]; pmce_stream_forward then serves W windows (int32 win[W,2], inclusive
 * [start,end], start == end repeats the frame) from those tables.  Results equal pmce_forward on the assembled windows.
 * Workspace: pmce_model_workspace_bytes(m, ceil(L/16)) for the precompute, (m, W) for the forward. */
int pmce_stream_precompute(pmce_model* m, const float* pose2d_frames, const float* feat_frames, int L, float* x0, float* gi0,
                           void* ws, size_t ws_bytes, pmce_stream_t stream);
int pmce_stream_forward(pmce_model* m, const float* x0, const float* gi0, const int* win, int W, int L, float* cam_mesh,
                        float* cam_pose, float* pose3d, float* pred_pose, void* ws, size_t ws_bytes, pmce_stream_t stream);
/* building blocks of the above */
int pmce_window_tokens_f32(const float* x0, const int* win, const float* tpos, const float* w2, const float* b2, float eps2,
                           float* X, float* XN, int W, int L, int T, int J, int C, pmce_stream_t stream);
/* XN written pre-split (see pmce_ln_chain_ex_f32). */
int pmce_window_tokens_ex_f32(const float* x0, const int* win, const float* tpos, const float* w2, const float* b2, float eps2,
                           float* X, float* XN, int W, int L, int T, int J, int C, int xn_split, pmce_stream_t stream);
int pmce_window_rows_f32(const float* src, const int* win, float* dst, int W, int L, int T, int ncols, pmce_stream_t stream);

/* enable != 0 (default): pmce_forward / pmce_decoder_forward run the image-feature branch (GRU, AdaLN parameters) and the
 * short joint-side kernels on a second, internally created HIP stream, forked from and joined back into the caller's
 * stream with events (hipGraph-capturable; results are identical).  0 keeps every launch on the caller's stream. */
int pmce_model_set_concurrency(pmce_model* m, int enable);
int pmce_model_get_concurrency(const pmce_model* m);
int pmce_model_get_split_overlap(const pmce_model* m);

/* Staggering of several forwards in flight (one handle per lane, shared weights): makes `stream` wait until the pose
 * lifter of the last pmce_forward enqueued on `m` has finished, so that the next batch's lifter (long matrix-bound GEMMs)
 * runs beside that batch's decoder (short latency-bound kernels) instead of beside its lifter.  No-op before m's first
 * forward.  Ordering only: results do not depend on it. */
int pmce_model_wait_lifter(pmce_model* m, pmce_stream_t stream);

/* Per-kernel-class timing of the forwards above (HIP events on the caller's stream).  enable != 0 starts
 * accumulating; pmce_model_profile_read synchronises the recorded events and returns, for class i, its
 * name, accumulated milliseconds and launch count; returns the number of classes. */
int pmce_model_profile(pmce_model* m, int enable);
int pmce_model_profile_read(pmce_model* m, int i, const char** name, double* ms, long long* launches);

/* ---------------------------------------------------------------------------------------------------------
 * Individual operators (what the model object launches; exported for the parity tests and for callers that
 * want one stage).  Shapes in comments; B = clips.
 * ------------------------------------------------------------------------------------------------------- */

/* C[m,n] = act(sum_k A[m,k] W[n,k] + bias[n]) + R[m,n]  — every nn.Linear / the packed GRU, AdaLN and upsample
 * products (timm Mlp/Attention Linear layers; CoevoDecoder.py:19-20,214-224).  K % 32 == 0.  act: 0 none,
 * 1 exact GELU.  Row maps: if a_div > 0 row r of A is at A + (r % a_div)*a_lo + (r / a_div)*a_hi, else r*lda;
 * same for C and R with c_*.  batch > 1 adds bs* element offsets per batch index (grid.z). */
int pmce_gemm_nt_f32(const float* A, const float* W, const float* bias, const float* R, float* C, int M, int N, int K,
                     long long lda, int ldw, long long ldc, int act, int a_div, long long a_lo, long long a_hi, int c_div,
                     long long c_lo, long long c_hi, int batch, long long bsA, long long bsW, long long bsBias,
                     long long bsC, pmce_stream_t stream);
/* Tuning aid only: force the tile configuration (0..3, -1 = automatic) and the persistent workgroups per CU (1..8, 0 =
 * automatic) of pmce_gemm_nt_f32 for the whole process (initial values: PMCE_GEMM_TILE / PMCE_GEMM_GRID, read once). */
int pmce_gemm_set_tuning(int tile, int grid_per_cu);
/* The same nn.Linear product on the f16 matrix pipe with fp32 operands, result and accuracy: W is split ONCE into f16
 * (hi, lo) planes of W[n] * 2^s(n) by pmce_gemm_pack_split_f16 (Wp: N*K floats of storage; wscale: N floats, 2^-s(n) - one power
 * of two per OUTPUT ROW n of W, chosen so that the row's largest |w| 2^s lies in [2^14, 2^15): an outlier row does not cost the
 * other rows' lo planes their bits), A is split into (hi, lo * 2^11) on the fly, C[m][n] = 2^-s(n) (Ahi Whi + Ahi Wlo + Alo Whi)
 * accumulated in fp32 - error at or below the fp32 product's own rounding.  A, bias, R, C stay fp32 and row-major (lda, ldc);
 * |A| must be below 65504 here (an element outside the f16 range yields inf/nan, never a silently wrong finite value; raw inputs
 * of arbitrary magnitude go through pmce_split_rows_scaled_f16 / pmce_gemm_nt_split_f16_rs below; inside a model call a non-finite
 * result also sets the model's overflow word, pmce_model_overflowed).
 * MI355X: while a kernel that issues f16 matrix instructions runs, packed-fp32 vector arithmetic (v_pk_{fma,mul,add}_f32) of
 * any other wave on the same CU may return wrong results (scripts/microbench/libpmce_diag.so: pmce_dbg_victim reproduces it).  No kernel of this library contains
 * such instructions, so its entries may overlap each other freely; do not overlap these entries with foreign kernels that do. */
int pmce_gemm_pack_split_f16(const float* W, int N, int K, int ldw, float* Wp, float* wscale, pmce_stream_t stream);
int pmce_gemm_nt_split_f16(const float* A, const float* Wp, const float* wscale, const float* bias, const float* R, float* C,
                           int M, int N, int K, long long lda, long long ldc, int act, int a_packed, pmce_stream_t stream);
/* c_packed != 0 (packed A, GELU, no residual, N % 32 == 0): the result is written pre-split as well - it is the A operand of the
 * next product (the lifter's fc1 -> fc2). */
int pmce_gemm_nt_split_f16_ex(const float* A, const float* Wp, const float* wscale, const float* bias, const float* R, float* C,
                              int M, int N, int K, long long lda, long long ldc, int act, int a_packed, int c_packed,
                              pmce_stream_t stream);
/* The same with mapped output rows: row r of C at C + (r % c_div)*c_lo + (r / c_div)*c_hi (the GRU layer-0 input projection
 * writes (b,t) rows time-major). */
int pmce_gemm_nt_split_f16_rowmap(const float* A, const float* Wp, const float* wscale, const float* bias, float* C, int M, int N,
                                  int K, long long lda, int c_div, long long c_lo, long long c_hi, pmce_stream_t stream);
/* RAW inputs of any finite fp32 magnitude: pmce_split_rows_scaled_f16 stores row m of A[M][lda] as f16 (hi, lo * 2^11) planes of
 * A[m] * 2^-e(m) (Ap: M*K floats of storage) and rscale[m] = 2^e(m), e(m) lifting the row's largest |a| into [2^14, 2^15) - the
 * per-row treatment pmce_gemm_pack_split_f16 gives W.  pmce_gemm_nt_split_f16_rs multiplies such an A:
 * C[m][n] = 2^e(m) 2^-s(n) (Ahi Whi + Ahi Wlo + Alo Whi) + bias[n]; c_div > 0 maps the output rows as in _rowmap (ldc == N then).
 * Rows holding inf / nan keep e = 0 and yield non-finite results in their own row only.
 * A row-scaled product needs K >= 128 (K = 32 .. 127 returns PMCE_ERR_ARG: the kernels read a tile's row scales after its last k-tile while
 * the LDS-DMA cursor runs up to three k-tile stages ahead; every row-scaled product of the path has K = 2048). */
int pmce_split_rows_scaled_f16(const float* A, long long M, int K, long long lda, float* Ap, float* rscale, pmce_stream_t stream);
int pmce_gemm_nt_split_f16_rs(const float* Ap, const float* rscale, const float* Wp, const float* wscale, const float* bias, float* C,
                              int M, int N, int K, long long ldc, int c_div, long long c_lo, long long c_hi, pmce_stream_t stream);
/* The BLOCKED weight layout (round 4; what the model packs and multiplies): pmce_gemm_pack_split_f16_blk writes the same planes as
 * pmce_gemm_pack_split_f16 as [ceil(N/64)][K/16][64 rows][16 hi | 16 lo] (Wp: ceil(N/64)*64*K floats of storage), so that what a tile
 * fetches per k-tile is contiguous 4 KB pieces instead of 64-byte pieces one weight row apart.  pmce_gemm_nt_split_f16_blk is every
 * product form above on such a weight: rscale != null -> A row-scaled (as _rs), c_div > 0 -> mapped output rows (ldc == N), else as
 * _ex.  Results are bit-identical to the row-major forms (same arithmetic, same k order). */
int pmce_gemm_pack_split_f16_blk(const float* W, int N, int K, int ldw, float* Wp, float* wscale, pmce_stream_t stream);
int pmce_gemm_nt_split_f16_blk(const float* A, const float* rscale, const float* Wblk, const float* wscale, const float* bias,
                               const float* R, float* C, int M, int N, int K, long long lda, long long ldc, int act, int a_packed,
                               int c_packed, int c_div, long long c_lo, long long c_hi, pmce_stream_t stream);
/* A product with N = 256 and a residual, followed by the LayerNorm chain of its consumer, in ONE launch (round 4; proj -> norm2 and
 * fc2 -> norm_s / norm_t -> next norm1 of a C = 256 lifter block, reference PoseEstimation.py:26-28,84-85,91-92,101-106): with
 * x = Ap W^T + bias + R,   y1 = ln1_w ? LN(x; ln1_w, ln1_b, ln1_eps) : x,   out1 = y1 (fp32 [M,256]; may alias R; may be NULL),
 * out2 = LN(y1; ln2_w, ln2_b, ln2_eps) written pre-split (the next product's packed A; may be NULL).  Ap pre-split [M,K]; Wp from
 * pmce_gemm_pack_split_f16 (w_blocked = 0) or _blk (1).  The same as pmce_gemm_nt_split_f16_blk + pmce_ln_chain_ex_f32 up to the
 * summation order of the row statistics (two-pass fp32 in both). */
int pmce_gemm_nt_split_f16_ln(const float* Ap, const float* Wp, int w_blocked, const float* wscale, const float* bias, const float* R,
                              int M, int K, const float* ln1_w, const float* ln1_b, float ln1_eps, float* out1, const float* ln2_w,
                              const float* ln2_b, float ln2_eps, float* out2, pmce_stream_t stream);
/* a_packed != 0: A is not fp32 but already split, [M][K/16][hi 16 f16 | lo*2^11 16 f16] (the layout the lifter's own
 * producers write; pmce_split_rows_f16 makes it from fp32 rows). */
int pmce_split_rows_f16(const float* A, long long M, int K, long long lda, float* Ap, pmce_stream_t stream);
/* Tuning aid only: force the tile configuration of pmce_gemm_nt_split_f16 (0: 128x256, 1: 128x128, 2: 64x128; -1 automatic). */
int pmce_gemm_split_set_tuning(int tile);
/* PoseEstimation.py:78-81 — x[tok] = joint_embed(pose2d) + imgfeat_embed(img_feat)[b,t] + spatial_pos[j]. */
int pmce_embed_tokens_f32(const float* pose2d, const float* E, const float* Wje, const float* bje, const float* spos,
                          float* x, long long ntok, int J, int C, pmce_stream_t stream);
/* The same followed by LayerNorm(w2, b2, eps2) of every token row (SpatialBlocks[0].norm1, PoseEstimation.py:13-29 via :83) in ONE launch:
 * x = the tokens (fp32 [ntok, C]), xn = their LayerNorm - fp32, or pre-split [row][C/16][16 hi | 16 lo*2^11] f16 when xn_split (the operand of a
 * three-product f16 GEMM).  Bit-identical to pmce_embed_tokens_f32 + pmce_ln_chain_ex_f32(out2).  C = 256 or 512. */
int pmce_embed_ln_f32(const float* pose2d, const float* E, const float* Wje, const float* bje, const float* spos, float* x,
                      long long ntok, int J, int C, const float* w2, const float* b2, float eps2, float* xn, int xn_split,
                      pmce_stream_t stream);
/* nn.LayerNorm chain over rows of C channels: y1 = (w1 ? LN(x;w1,b1,eps1) : x) + add[(row/add_div)%add_mod];
 * out1 = y1 (optional); out2 = LN(y1;w2,b2,eps2) (optional).  norm1/norm2/norm_s/norm_t (PoseEstimation.py:17,23,58-59). */
/* _ex forms: out2 / out / XN written pre-split ([row][C/16][16 hi | 16 lo*2^11] f16 in the bytes of the fp32 row), i.e. directly as
 * the A operand of pmce_gemm_nt_split_f16_ex(a_packed = 1). */
int pmce_ln_chain_ex_f32(const float* x, long long rows, int C, const float* w1, const float* b1, float eps1, const float* add,
                         int add_div, int add_mod, float* out1, const float* w2, const float* b2, float eps2, float* out2,
                         int out2_split, pmce_stream_t stream);
int pmce_seq_attention_ex_f32(const float* qkv, float* out, int nseq, int N, int C, int seq_div, long long seq_lo,
                              long long seq_hi, long long tok_stride, int out_split, pmce_stream_t stream);
/* The same attention (timm Attention of PoseEstimation.py:78-104) on the f16 matrix pipe in the three-product form: q, k, v read
 * as fp32 (split into f16 (hi, lo) planes inside the kernel), the result WRITTEN pre-split ([row][C/16][16 hi | 16 lo*2^11] f16 in
 * the bytes of the fp32 row) as the A operand of proj.  Sequence / token addressing as pmce_seq_attention_f32.  C = 256 or 512 (8 heads),
 * N = 16, 17 or 19 (pmce_seq_attention_split_supported tells); anything else returns PMCE_ERR_ARG - callers keep fp32 qkv and
 * pmce_seq_attention_ex_f32 there.  A non-finite result sets the calling thread's overflow sink like the products do. */
int pmce_seq_attention_split_supported(int N, int C);
int pmce_seq_attention_split_f16(const float* qkv, float* out_planes, int nseq, int N, int C, int seq_div, long long seq_lo,
                                 long long seq_hi, long long tok_stride, pmce_stream_t stream);
int pmce_ln_chain_f32(const float* x, long long rows, int C, const float* w1, const float* b1, float eps1,
                      const float* add, int add_div, int add_mod, float* out1, const float* w2, const float* b2,
                      float eps2, float* out2, pmce_stream_t stream);
/* timm Attention core on qkv[tok][3C] for sequences of N <= 32 tokens, 8 heads (PoseEstimation.py:19 / CoevoDecoder.py:118-131).
 * sequence s, position i -> token (s % seq_div)*seq_lo + (s / seq_div)*seq_hi + i*tok_stride. */
int pmce_seq_attention_f32(const float* qkv, float* out, int nseq, int N, int C, int seq_div, long long seq_lo,
                           long long seq_hi, long long tok_stride, pmce_stream_t stream);
/* PoseEstimation.py:62-66,109-113 — LayerNorm(1e-5) + Linear(C->3) + Conv2d(T->1) frame fusion. */
int pmce_lifter_head_f32(const float* x, const float* lnw, const float* lnb, const float* Wr, const float* br,
                         const float* wf, const float* bf, float* pose3d, int B, int T, int J, int C, pmce_stream_t stream);
/* prew != NULL: x is the LAST TemporalBlock's output BEFORE its post-norm, and every row passes through LayerNorm(prew, preb, pre_eps) (norm_t,
 * PoseEstimation.py:92) on the way in - bit-identical to pmce_ln_chain_f32(out1) followed by pmce_lifter_head_f32, without the launch and the round trip. */
int pmce_lifter_head_ex_f32(const float* x, const float* prew, const float* preb, float pre_eps, const float* lnw, const float* lnb,
                            const float* Wr, const float* br, const float* wf, const float* bf, float* pose3d, int B, int T, int J, int C,
                            pmce_stream_t stream);

/* Fused nn.GRU time step for ndir directions: gh = h_prev W_hh^T + b_hh on the matrix cores, then the gate update
 * (CoevoDecoder.py:216-221); gi = W_ih x + b_ih comes from pmce_gemm_nt_f32.  hp == NULL means h_prev = 0. */
int pmce_gru_step_f32(const float* gi0, const float* gi1, const float* whh0, const float* whh1, const float* bhh0,
                      const float* bhh1, const float* hp0, const float* hp1, float* ho0, float* ho1, long long gi_rs,
                      long long h_rs, int B, int H, int ndir, pmce_stream_t stream);
/* The same step with gh = h W_hh^T in the three-product f16 form (pmce_gemm_nt_split_f16's arithmetic): whh0p / whh1p are rows of
 * ONE weight packed by pmce_gemm_pack_split_f16 (K = H), wscale its scale pair. */
int pmce_gru_step_split_f32(const float* gi0, const float* gi1, const float* whh0p, const float* whh1p, const float* wscale,
                            const float* bhh0, const float* bhh1, const float* hp0, const float* hp1, float* ho0, float* ho1,
                            long long gi_rs, long long h_rs, int B, int H, int ndir, pmce_stream_t stream);
/* The same on a W_hh packed by pmce_gemm_pack_split_f16_blk (the blocked layout: the 16 rows a DMA instruction fetches are 1 KB contiguous;
 * whh0b / whh1b = the first row of each direction, a multiple of 64 rows apart): what the model runs.  B <= 64 runs a small-batch kernel (a
 * workgroup per 8 hidden units: 256 workgroups whatever the batch), larger batches 64 rows x 32 units per workgroup - same numbers, bit for
 * bit, in both layouts and at every batch size. */
int pmce_gru_step_split_blk_f32(const float* gi0, const float* gi1, const float* whh0b, const float* whh1b, const float* wscale,
                                const float* bhh0, const float* bhh1, const float* hp0, const float* hp1, float* ho0, float* ho1,
                                long long gi_rs, long long h_rs, int B, int H, int ndir, pmce_stream_t stream);
/* y = x / denom (PMCE.py:18). */
int pmce_div_scalar_f32(const float* x, float* y, long long n, float denom, pmce_stream_t stream);

/* CoevoDecoder.py:232 — vertxs[b][v][:] = joints[b][vj[v]][:]; vj int32[431].  Bit-exact copy. */
int pmce_vertex_init_gather_f32(const float* joints, const int* vj, float* vt, int B, int J, pmce_stream_t stream);
/* Key/value side of the vertex<-joint CrossAttention (CoevoDecoder.py:47-62,83) with Wq / proj folded in:
 * GB[b][inst*128 + (gamma 0..63 | beta 64..127)] are the AdaLN parameters; Kf,Vf [B,64,64], s0 [B,64]. */
int pmce_ca_fold_f32(const float* xk, const float* xv, const float* GB, int gb_stride, int iq, int ik, int iv,
                     const float* Wq, const float* bq, const float* Wk, const float* bk, const float* Wv, const float* bv,
                     const float* Wp, float* Kf, float* s0, float* Vf, int B, int J, pmce_stream_t stream);
/* The same, also (img != NULL, J <= 23) writing the operands as the LDS image pmce_vertex_ca_mlp_pk_f32's split_f16 form copies: per clip
 * pmce_ca_image_floats() floats - s0, two scales, Kf and Vf as (hi | lo) f16 fragment planes scaled by one power of two each.  Kf / s0 / Vf may
 * each be NULL when only the image is wanted. */
int pmce_ca_image_floats(void);
int pmce_ca_fold_img_f32(const float* xk, const float* xv, const float* GB, int gb_stride, int iq, int ik, int iv,
                         const float* Wq, const float* bq, const float* Wk, const float* bk, const float* Wv, const float* bv,
                         const float* Wp, float* Kf, float* s0, float* Vf, float* img, int B, int J, pmce_stream_t stream);
/* The joint side of a CoevoBlock in ONE launch (what the model runs): the joint embedding of CoevoDecoder.py:177-180,184 -
 * jf = joint_proj(jt) + joint_pos_embed, xk = proj_j2v_dim(jf) + j2v_K_embed, xv = jf - followed by the fold above.  jf_out [B,J,64] may be
 * NULL (only the block-3 joint stream reads it). */
int pmce_joint_prep_f32(const float* jt, const float* Wj, const float* bj, const float* jpos, const float* Wj2v, const float* bj2v,
                        const float* j2vK, float* jf_out, const float* GB, int gb_stride, int iq, int ik, int iv, const float* Wq,
                        const float* bq, const float* Wk, const float* bk, const float* Wv, const float* bv, const float* Wp,
                        float* Kf, float* s0, float* Vf, float* img, int B, int J, pmce_stream_t stream);
/* THE north-star kernel: fused AdaLN + vertex<-joint cross-attention + residual for 431 query tokens
 * (first line of CrossAttentionBlock.forward, CoevoDecoder.py:83).  xq[B,431,64], or xq == NULL and
 * xq := Wv3*vt + Eq formed on the fly from vt[B,431,3]. */
int pmce_vertex_ca_f32(const float* xq, const float* vt, const float* Wv3, const float* Eq, const float* Kf,
                       const float* s0, const float* Vf, const float* bp, float* out, int B, int J, pmce_stream_t stream);
/* x + Mlp(AdaLN(x)) on [B,431,64] (CoevoDecoder.py:85-86,104); optional Linear(64->3)+coordinate residual (:189). */
int pmce_adaln_mlp_f32(const float* xin, const float* GB, int gb_stride, int inst, const float* W1, const float* b1,
                       const float* W2, const float* b2, float* yout, const float* Wc, const float* bc, const float* vt_in,
                       float* vt_out, int B, pmce_stream_t stream);

/* The whole CrossAttentionBlock of the vertex stream in one launch (reference CoevoDecoder.py:82-87): vertex_ca followed by
 * its FFN (adaln_mlp with AdaLN instance `inst` of GB), the intermediate never leaving registers; arguments as those two.
 * Same arithmetic in the same order as pmce_vertex_ca_f32 + pmce_adaln_mlp_f32.  J <= 23 runs fused; beyond that one clip's
 * folded operands do not fit beside the FFN weights in LDS and the call runs the two kernels through `scratch` [B,431,64]
 * (may be NULL when J <= 23). */
int pmce_vertex_ca_mlp_f32(const float* xq, const float* vt, const float* Wv3, const float* Eq, const float* Kf,
                           const float* s0, const float* Vf, const float* bp, const float* GB, int gb_stride, int inst,
                           const float* W1, const float* b1, const float* W2, const float* b2, float* yout, float* scratch,
                           int B, int J, pmce_stream_t stream);
/* _pk forms of the two FFN-carrying kernels.  split_f16 != 0 runs the 64->256->64 FFN in the three-product f16 form (activations split in
 * registers; fp32 accumulate, fp32 results); ffn_img != NULL: every workgroup COPIES the FFN's f16 planes into LDS (LDS-DMA) from an image
 * made once (pmce_ffn_pack_f16 from the same W1 [256,64] / W2 [64,256]: pmce_ffn_image_floats() floats, 16-byte aligned) instead of converting
 * W1 / W2 itself - the same bits; pmce_model_finalize makes the six images.  pmce_vertex_ca_mlp_pk_f32 with split_f16 != 0 also runs the
 * cross-attention's two contractions in that form and REQUIRES ca_img = the folded operands' image of pmce_ca_fold_img_f32 /
 * pmce_joint_prep_f32 (Kf / s0 / Vf are then not read); J > 23 takes the two-launch form through `scratch` (fp32 attention from Kf / s0 / Vf). */
int pmce_ffn_image_floats(void);
int pmce_ffn_pack_f16(const float* W1, const float* W2, float* ffn_img, pmce_stream_t stream);
int pmce_adaln_mlp_pk_f32(const float* xin, const float* GB, int gb_stride, int inst, const float* W1, const float* b1,
                          const float* W2, const float* b2, float* yout, const float* Wc, const float* bc, const float* vt_in,
                          float* vt_out, int B, int split_f16, const float* ffn_img, pmce_stream_t stream);
int pmce_vertex_ca_mlp_pk_f32(const float* xq, const float* vt, const float* Wv3, const float* Eq, const float* Kf,
                              const float* s0, const float* Vf, const float* bp, const float* GB, int gb_stride, int inst,
                              const float* W1, const float* b1, const float* W2, const float* b2, float* yout, float* scratch,
                              int B, int J, int split_f16, const float* ffn_img, const float* ca_img, pmce_stream_t stream);

/* fp32 pipe: qkv = Linear(64->192)(AdaLN(x)) on [B,431,64] (CoevoDecoder.py:103,120), then
 * y = x + proj(softmax(q k^T/sqrt(32)) v), 2 heads, 431x431 per clip (CoevoDecoder.py:118-131,103). */
int pmce_adaln_qkv_f32(const float* xin, const float* GB, int gb_stride, int inst, const float* Wqkv, const float* bqkv,
                       float* qkv, int B, pmce_stream_t stream);
int pmce_vertex_sa_f32(const float* xin, const float* qkv, const float* Wp, const float* bp, float* yout, int B,
                       pmce_stream_t stream);
/* The qkv weight [192,64] as the f16 image pmce_vertex_sab_split_f32 copies into LDS: pmce_qkv_image_floats() floats, 16-byte aligned, made
 * once per weight. */
int pmce_qkv_image_floats(void);
int pmce_qkv_pack_f16(const float* Wqkv, float* qkv_img, pmce_stream_t stream);
/* The attention half of the vertex stream's AdaLN Block in ONE launch, three-product f16 form (what a model in split_f16 mode runs):
 *   y = x + proj(softmax(q k^T/sqrt(32)) v),  [q|k|v] = Linear(64->192)(AdaLN(x))   (CoevoDecoder.py:103,118-131)
 * without a [B,431,192] fp32 QKV round trip: q, k, v come out of the qkv product's accumulators in the attention's own operand
 * layouts (every contraction but the output projection as three f16 matrix products of (hi, lo) halves, fp32 accumulate); k and v pass through `scratch`
 * (pmce_vertex_sab_scratch_floats(B) floats, 16-byte aligned, caller-owned, contents meaningless outside the call) as f16 (hi | lo)
 * fragment planes.  qkv_img = pmce_qkv_pack_f16(Wqkv).  B > 128: one workgroup per clip; otherwise two. */
long long pmce_vertex_sab_scratch_floats(int B);
int pmce_vertex_sab_split_f32(const float* xin, const float* GB, int gb_stride, int inst, const float* qkv_img, const float* bqkv,
                              const float* Wp, const float* bp, float* scratch, float* yout, int B, pmce_stream_t stream);
/* k|v of the joint<-vertex CrossAttention for the 431 vertex tokens: kv[B,431,128] (CoevoDecoder.py:52-53,83,183). */
int pmce_tokens_kv_f32(const float* xk, const float* xv, const float* vt, const float* Wv3, const float* Ev,
                       const float* Wv2j, const float* Ek, const float* GB, int gb_stride, int ik, int iv, const float* Wk,
                       const float* bk, const float* Wv, const float* bv, float* kv, int B, pmce_stream_t stream);
/* The same with its three 64 x 64 products (proj_v2j_dim, wk, wv) in the three-product f16 form (what a model in split_f16 mode runs): tkv_img =
 * pmce_tkv_pack_f16(Wv2j, Wk, Wv), pmce_tkv_image_floats() floats, 16-byte aligned, made once; NULL = the fp32 form above. */
int pmce_tkv_image_floats(void);
int pmce_tkv_pack_f16(const float* Wv2j, const float* Wk, const float* Wv, float* tkv_img, pmce_stream_t stream);
int pmce_tokens_kv_pk_f32(const float* xk, const float* xv, const float* vt, const float* Wv3, const float* Ev,
                          const float* Wv2j, const float* Ek, const float* GB, int gb_stride, int ik, int iv, const float* Wk,
                          const float* bk, const float* Wv, const float* bv, float* kv, int B, const float* tkv_img, pmce_stream_t stream);
/* Joint stream of a CoevoBlock (CoevoDecoder.py:183,187,189): stage 1 = joint<-vertex cross-attention +
 * residual only, 2 = + FFN, 3 = + self-attention block + coordinate head.  wptr: 18 weight pointers
 * (wq,bq,proj_w,proj_b,fc1_w,fc1_b,fc2_w,fc2_b,qkv_w,qkv_b,sproj_w,sproj_b,sfc1_w,sfc1_b,sfc2_w,sfc2_b,coor_w,coor_b);
 * inst: AdaLN instance ids (normq, norm2, SA.norm1, SA.norm2). */
int pmce_joint_stream_f32(const float* xq, const float* jQ, const float* kv, const float* GB, int gb_stride,
                          const float* const* wptr, const int* inst, const float* jt, float* y_out, float* pose_out, int B,
                          int J, int stage, pmce_stream_t stream);
/* Operand of the packed upsample+residual product: A[b] = [relu(g[b]) | vt[b] flattened | 0-pad] (CoevoDecoder.py:238-244). */
int pmce_build_final_operand_f32(const float* g, const float* vt, float* A, int B, int KP, pmce_stream_t stream);
/* packed != 0 (KP % 16 == 0): the rows are written pre-split - the A operand of pmce_gemm_nt_split_f16*(a_packed = 1); what the model runs. */
int pmce_build_final_operand_pk_f32(const float* g, const float* vt, float* A, int B, int KP, int packed, pmce_stream_t stream);
/* lib/core/base.py:223-225 — out[b][r][:] = sum_nz data * (mesh[b][col][:] * scale); CSR regressor [R,6890]. */
int pmce_j_regress_f32(const float* mesh, const int* indptr, const int* indices, const float* data, float* out, int B,
                       int R, int NVF, float scale, pmce_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Evaluation metrics directly behind the path: replace the per-batch D2H + numpy of
 * compute_both_err (data/PW3D/dataset.py:269-282) and the per-sample loop of evaluate (:351-462).
 * ------------------------------------------------------------------------------------------------------- */
/* Per sample b: mpvpe[b] = mean_v ||(pm*scale - rp) - (gm*scale - rg)||; joints P = pj - rowsum*rp (if rowsum), minus
 * joint root_j, restricted to eval_idx (int32[n_eval]); mpjpe[b] = mean ||P-G||; pampjpe[b] after rigid_align
 * (lib/coord_utils.py:151-173, fp64).  rp/rg NULL -> mesh roots are the samples' own joint root_j (compute_both_err).
 * out_pe/out_ge (optional, [B,n_eval,3]) receive the aligned eval joints for pmce_accel_error_f32.
 * V == 0 (pm, gm, rp, rg, rowsum NULL): joints only - the pose-only flavours compute_joint_err / evaluate_joint
 * (data/Human36M/dataset.py:600-713: root 0, the 14 eval joints; data/PW3D/dataset.py:260-349: COCO set, root = joint J-2, every joint)
 * and MPII3D.evaluate (data/MPII3D/dataset.py:539-624: root 0, all 17 joints); mpvpe[b] = 0 then. */
int pmce_sample_errors_f32(const float* pm, const float* gm, float scale, int V, const float* rp, const float* rg,
                           const float* pj, const float* gj, int NJ, const float* rowsum, const int* eval_idx, int n_eval,
                           int root_j, float* out_mpvpe, float* out_mpjpe, float* out_pampjpe, float* out_pe, float* out_ge,
                           int B, pmce_stream_t stream);
/* Per-sample acceleration error (lib/coord_utils.py:218-245 as used at data/PW3D/dataset.py:415-429): 0 for the first and
 * last sample of each sequence (seq int32[N], samples of a sequence contiguous). */
int pmce_accel_error_f32(const float* pe, const float* ge, const int* seq, float* out, int N, int n_eval,
                         pmce_stream_t stream);

/* Sliding-window clip assembly on the GPU (lib/_img_utils.py:42-55; demo lib/utils/_dataset_demo.py:98-102):
 * per-frame tables pose[L,J,2], feat[L,2048] + windows int32[W,2] (inclusive [start,end]; start == end repeats the frame)
 * -> out_pose[W,16,J,2], out_feat[W,16,2048]. */
int pmce_assemble_windows_f32(const float* pose, const float* feat, const int* win, float* out_pose, float* out_feat, int W,
                              int L, int J, pmce_stream_t stream);

/* Detector keypoints -> model input, per frame (data/PW3D/dataset.py:185-204 add_pelvis_and_neck /
 * normalize_screen_coordinates, applied at :160-161 and :235-237): kp[L,J0,kp_stride] pixel coordinates (x, y first),
 * shape int32[L,2] = (height, width) -> out[L,J0+n_extra,2]; n_extra = 1 appends the pelvis (hip midpoint), 2 also the
 * neck (shoulder midpoint); x' = x/w*2 - 1, y' = y/w*2 - h/w. */
int pmce_prepare_pose2d_f32(const float* kp, int kp_stride, const int* shape, float* out, int L, int J0, int n_extra, int lhip,
                            int rhip, int lsho, int rsho, pmce_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PMCE_HIP_H */
