"""Wave-specialised split-f16 GEMM (gemm_split_ws.hip) against the 4-wave kernel: bitwise equality, run-to-run
reproducibility, hand-off timeouts and launch time at the lifter's shapes."""
import sys, torch
sys.path.insert(0, ".")
from pmce_amd import ops, _lib
lib = _lib.load()
dev = "cuda"
torch.manual_seed(0)

def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

quick = "--quick" in sys.argv
shapes = [  # M, N, K, act, res, c_packed
    (69632, 1536, 512, 0, False, False), (69632, 512, 512, 0, True, False), (69632, 1024, 512, 1, False, True),
    (69632, 512, 1024, 0, True, False), (69632, 768, 256, 0, False, False), (69632, 256, 256, 0, True, False),
    (69632, 512, 256, 1, False, True), (69632, 256, 512, 0, True, False),
    (40000, 640, 256, 1, False, False), (17408, 1536, 512, 0, False, False), (50001, 1024, 128, 0, True, False),
]
for (M, N, K, act, res, cpk) in shapes:
    A = torch.randn(M, K, device=dev); A[::7] *= 1e-3
    W = torch.randn(N, K, device=dev) * 0.05
    b = torch.randn(N, device=dev)
    R = torch.randn(M, N, device=dev) if res else None
    Wp, ws = ops.pack_split_f16(W)
    Ap = ops.split_rows_f16(A)
    run = lambda: ops.gemm_nt_split(Ap, Wp, ws, b, R, act, a_packed=True, c_packed=cpk)
    lib.pmce_gemm_split_set_tuning(0); c_old = run().clone()
    lib.pmce_gemm_split_set_tuning(3); c_ws = run().clone()
    same = torch.equal(c_old.view(torch.int32), c_ws.view(torch.int32))
    nbad = 0 if same else int((c_old.view(torch.int32) != c_ws.view(torch.int32)).sum().item())
    rep_bad = 0
    for _ in range(5 if quick else 30):
        c2 = run()
        rep_bad += int(not torch.equal(c2.view(torch.int32), c_ws.view(torch.int32)))
    t_ws = timeit(run)
    lib.pmce_gemm_split_set_tuning(0); t_old = timeit(run)
    lib.pmce_gemm_split_set_tuning(-1); t_auto = timeit(run)
    fl = 3 * 2.0 * M * N * K
    print(f"{M:6d} x {N:5d} x {K:5d} act={act} res={int(res)} cpk={int(cpk)}: 4-wave {t_old:7.1f} us  ws {t_ws:7.1f} us ({fl/t_ws/1e6:6.0f} TF issued = {fl/t_ws/1e6/2500:.3f})  auto {t_auto:7.1f} us | "
          f"bitwise {'equal' if same else f'DIFFERENT ({nbad})'}  reruns differing {rep_bad}  timeouts {lib.pmce_gemm_ws_timeouts()}", flush=True)
lib.pmce_gemm_split_set_tuning(-1)
