"""Bystander test: does a kernel on one stream change its results because a matrix-pipe kernel runs on another?

Self-checking bystanders (scripts/microbench/csrc/dbg_victims.hip: every wave recomputes one fixed function of its own registers 400 times and
counts iterations that differ from the first) next to (a) matrix-pipe spinners - one MFMA shape on register operands, no memory
traffic - and (b) the real GEMMs of this path; plus the lifter head kernel (compiler-scheduled, 272 workgroups) against its own
stand-alone output.  On MI355X the f16 matrix instructions disturb packed-fp32 (op_sel forms) arithmetic of bystanders; the
fp32 matrix instructions do not.  That is why the split-f16 mode runs everything on one stream (csrc/model.cpp two_streams)."""
import ctypes as C, sys, torch
sys.path.insert(0, ".")
from pmce_amd import ops, _lib
from scripts.microbench import diag      # bystander / spinner kernels: the diagnostics library
lib = _lib.load()
dlib = diag.load()
dev = "cuda"
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)
tab = ((torch.arange(4096 * 1024 + 32 * 1024, device=dev, dtype=torch.int64) % 8191).float() * 0.5).contiguous()
sink = torch.zeros(256, device=dev)
shapes = ["32x32x16 f16", "16x16x32 f16", "32x32x8 f16 (CDNA3 shape)", "32x32x16 bf16", "32x32x2 f32", "32x32x16 fp8", "16x16x16 f16 (CDNA3 shape)"]
vict = {6: "packed fp32, op_sel forms", 0: "packed fp32, compiler-chosen", 1: "plain fp32", 2: "butterfly sums (ds_bpermute)", 3: "fp32 matrix pipe",
        7: "global loads"}
M, N, K = 4096, 3072, 2048
A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev)
Wp, ws = ops.pack_split_f16(W)
outg = torch.empty(M, N, device=dev)
def gemm_split(m=M):
    _lib.check(lib.pmce_gemm_nt_split_f16(_lib.ptr(A), _lib.ptr(Wp), _lib.ptr(ws), _lib.ptr(b), None, _lib.ptr(outg), m, N, K, K, N, 0, 0,
                                          C.c_void_p(s2.cuda_stream)))
def gemm_f32(m=M):
    _lib.check(lib.pmce_gemm_nt_f32(_lib.ptr(A), _lib.ptr(W), _lib.ptr(b), None, _lib.ptr(outg), m, N, K, K, K, N, 0, 0, 0, 0, 0, 0, 0, 1,
                                    0, 0, 0, 0, C.c_void_p(s2.cuda_stream)))
def trial(aggr, kind, reps=10):
    bad = torch.zeros(4, dtype=torch.int32, device=dev)
    for r in range(reps):
        aggr()
        diag.check(dlib.pmce_dbg_victim(kind, _lib.ptr(bad), 1024, 400, _lib.ptr(tab), C.c_void_p(s1.cuda_stream)))
        torch.cuda.synchronize()
    return bad.tolist()[0]
rows = [("nothing", lambda: None), ("fp32 GEMM of this path", lambda: [gemm_f32() for _ in range(3)]),
        ("split-f16 GEMM of this path", lambda: [gemm_split() for _ in range(3)])]
rows += [(f"spinner {n}", (lambda k: lambda: diag.check(dlib.pmce_dbg_mfma_spin(k, _lib.ptr(sink), 512, 20000, C.c_void_p(s2.cuda_stream))))(k))
         for k, n in enumerate(shapes)]
print("lanes (of 262144) whose result changed at least once, per bystander kind:")
for label, aggr in rows:
    print(f"  next to {label:30s}:", {v: trial(aggr, k) for k, v in vict.items()}, flush=True)

# the lifter head next to the small-M GEMMs that overlap it in a two-stream forward
B, T, J, Cc = 64, 16, 17, 256
X = torch.randn(B * T * J, Cc, device=dev)
lnw = torch.randn(Cc, device=dev); lnb = torch.randn(Cc, device=dev)
Wr = torch.randn(3, Cc, device=dev); br = torch.randn(3, device=dev); wf = torch.randn(T, device=dev); bf = torch.randn(1, device=dev)
def head():
    out = torch.empty(B, J, 3, device=dev)
    _lib.check(lib.pmce_lifter_head_f32(_lib.ptr(X), _lib.ptr(lnw), _lib.ptr(lnb), _lib.ptr(Wr), _lib.ptr(br), _lib.ptr(wf), _lib.ptr(bf),
                                        _lib.ptr(out), B, T, J, Cc, C.c_void_p(s1.cuda_stream)))
    return out
ref = head(); torch.cuda.synchronize()
for label, aggr in (("nothing", lambda: None), ("fp32 GEMM 64x3072x2048", lambda: gemm_f32(64)), ("split-f16 GEMM 64x3072x2048", lambda: gemm_split(64))):
    bad = 0
    for r in range(200):
        aggr(); aggr()
        outs = [head() for _ in range(6)]
        torch.cuda.synchronize()
        bad += any(not torch.equal(o, ref) for o in outs)
    print(f"lifter head next to {label:28s}: {bad}/200 trials with an output that differs from the stand-alone run", flush=True)
