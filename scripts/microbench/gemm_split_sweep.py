"""Split-f16 GEMM: launch time per tile configuration / persistent-grid size (tuning knobs of pmce_gemm_set_tuning)."""
import sys, torch
sys.path.insert(0, ".")
from pmce_amd import ops, _lib
lib = _lib.load()
dev = "cuda"
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
M = 69632
for (N, K, act, res) in [(512, 512, 0, True), (1536, 512, 0, False), (1024, 512, 1, False), (512, 1024, 0, True)]:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
    R = torch.randn(M, N, device=dev) if res else None
    Wp, ws = ops.pack_split_f16(W)
    out = []
    Ap = ops.split_rows_f16(A)
    for tile in (0, 1, 2):
        lib.pmce_gemm_split_set_tuning(tile)
        t = timeit(lambda: ops.gemm_nt_split(A, Wp, ws, b, R, act))
        tp = timeit(lambda: ops.gemm_nt_split(Ap, Wp, ws, b, R, act, a_packed=True))
        out.append(f"t{tile}:{t*1e3:.0f}/{tp*1e3:.0f}")
    lib.pmce_gemm_split_set_tuning(-1)
    print(f"N={N} K={K} act={act} res={int(res)} us: " + " ".join(out), flush=True)
