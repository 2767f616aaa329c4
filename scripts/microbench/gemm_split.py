"""Split-f16 GEMM vs fp32-MFMA GEMM: error against an fp64 product and launch time, at the lifter's shapes."""
import sys, torch
sys.path.insert(0, ".")
from pmce_amd import ops
torch.manual_seed(0)
dev = "cuda"

def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

M = int(sys.argv[1]) if len(sys.argv) > 1 else 69632
for (N, K, act, res) in [(512, 512, 0, True), (1536, 512, 0, False), (1024, 512, 1, False), (512, 1024, 0, True),
                         (256, 256, 0, True), (768, 256, 0, False), (512, 256, 1, False), (256, 512, 0, True)]:
    A = torch.randn(M, K, device=dev)
    A[::7] *= 1e-3   # some small rows
    W = torch.randn(N, K, device=dev) * 0.05
    b = torch.randn(N, device=dev)
    R = torch.randn(M, N, device=dev) if res else None
    Wp, ws = ops.pack_split_f16(W)
    c32 = ops.gemm_nt(A, W, b, R, act)
    csp = ops.gemm_nt_split(A, Wp, ws, b, R, act)
    Ap = ops.split_rows_f16(A)
    cpk = ops.gemm_nt_split(Ap, Wp, ws, b, R, act, a_packed=True)
    ms = min(M, 4096)
    ref = A[:ms].double() @ W.double().T + b.double()
    if act: ref = torch.nn.functional.gelu(ref)
    if res: ref = ref + R[:ms].double()
    e32 = (c32[:ms].double() - ref).abs().max().item(); esp = (csp[:ms].double() - ref).abs().max().item(); epk = (cpk[:ms].double() - ref).abs().max().item()
    t32 = timeit(lambda: ops.gemm_nt(A, W, b, R, act)); tsp = timeit(lambda: ops.gemm_nt_split(A, Wp, ws, b, R, act)); tpk = timeit(lambda: ops.gemm_nt_split(Ap, Wp, ws, b, R, act, a_packed=True))
    fl = 2.0 * M * N * K
    by = 4.0 * (M * K + M * N * (2 if res else 1) + N * K)
    print(f"M={M} N={N} K={K} act={act} res={int(res)}: err fp32 {e32:.2e} split {esp:.2e} (|ref|max {ref.abs().max().item():.1f}) | "
          f"fp32 {t32*1e3:.0f} us {fl/t32/1e9:.0f} TF | split {tsp*1e3:.0f} us {fl/tsp/1e9:.0f} TF-equiv {by/tsp/1e6:.0f} GB/s | packed-A err {epk:.2e} {tpk*1e3:.0f} us {fl/tpk/1e9:.0f} TF-equiv {by/tpk/1e6:.0f} GB/s", flush=True)
