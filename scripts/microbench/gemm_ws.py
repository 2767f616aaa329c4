"""Wave-specialised split-f16 GEMM (gemm_split_ws.hip) against the 4-wave kernel: bitwise equality, run-to-run
reproducibility, hand-off timeouts and launch time at the lifter's shapes."""
import sys, torch
sys.path.insert(0, ".")
from pmce_amd import ops, _lib
from scripts.microbench import diag      # the wave-specialised kernel is in the diagnostics library (not in the product)
lib = _lib.load()
dlib = diag.load()
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
for (M, N, K, act, res, cpk) in ([] if ("--ablate" in sys.argv or "--timeline" in sys.argv or "--check-opt" in sys.argv) else shapes):
    A = torch.randn(M, K, device=dev); A[::7] *= 1e-3
    W = torch.randn(N, K, device=dev) * 0.05
    b = torch.randn(N, device=dev)
    R = torch.randn(M, N, device=dev) if res else None
    Wp, ws = ops.pack_split_f16(W)
    Ap = ops.split_rows_f16(A)
    run_old = lambda: ops.gemm_nt_split(Ap, Wp, ws, b, R, act, a_packed=True, c_packed=cpk)
    run = lambda: diag.gemm_nt_split(0, Ap, Wp, ws, b, R, act, a_packed=True, c_packed=cpk)
    lib.pmce_gemm_split_set_tuning(0); c_old = run_old().clone()
    c_ws = run().clone()
    same = torch.equal(c_old.view(torch.int32), c_ws.view(torch.int32))
    nbad, where = 0, ""
    if not same:
        bad = c_old.view(torch.int32) != c_ws.view(torch.int32)
        nbad = int(bad.sum().item())
        rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
        where = (f" rows {int(rows[0])}..{int(rows[-1])} ({rows.numel()} rows) cols {int(cols[0])}..{int(cols[-1])} ({cols.numel()} cols) "
                 f"max|d| {(c_old.float() - c_ws.float()).abs().nan_to_num(1e30).max().item():.3g} row%192 hist {torch.bincount(rows % 192 // 32, minlength=6).tolist()}")
    rep_bad = 0
    for _ in range(5 if quick else 30):
        c2 = run()
        rep_bad += int(not torch.equal(c2.view(torch.int32), c_ws.view(torch.int32)))
    t_ws = timeit(run)
    lib.pmce_gemm_split_set_tuning(0); t_old = timeit(run_old)
    lib.pmce_gemm_split_set_tuning(-1); t_auto = timeit(run_old)
    fl = 3 * 2.0 * M * N * K
    print(f"{M:6d} x {N:5d} x {K:5d} act={act} res={int(res)} cpk={int(cpk)}: 4-wave {t_old:7.1f} us  ws {t_ws:7.1f} us ({fl/t_ws/1e6:6.0f} TF issued = {fl/t_ws/1e6/2500:.3f})  auto {t_auto:7.1f} us | "
          f"bitwise {'equal' if same else f'DIFFERENT ({nbad}){where}'}  reruns differing {rep_bad}  timeouts {dlib.pmce_gemm_ws_timeouts()}", flush=True)
lib.pmce_gemm_split_set_tuning(-1)

if "--ablate" in sys.argv:  # needs a library built with PMCE_EXTRA_HIPCC_FLAGS=-DPMCE_WS_ABLATE
    import ctypes
    raw = ctypes.CDLL(diag.LIB_PATH)
    names = {0: "full", 1: "no stores", 2: "no DMA", 3: "no stores, no DMA", 4: "no MFMA", 5: "no MFMA, no stores", 6: "no MFMA, no DMA",
             8: "DMA from a 4 KB window", 9: "4 KB window, no stores", 18: "no DMA, no land wait", 19: "no DMA, no land wait, no stores",
             7: "loop only (no MFMA/DMA/stores)", 23: "loop only, no land wait"}
    names.update({256: "62 of 64 stores per wave and tile", 512: "56 of 64 stores", 32: "A from a 4 KB window", 64: "W from a 4 KB window", 33: "A window, no stores", 65: "W window, no stores"})
    for o in (0, 1, 8, 9):
        names[200 + o] = f"OPT={o}"
        names[300 + o] = f"OPT={o} no DMA/stores"
    lib.pmce_gemm_split_set_tuning(3)
    for (M, N, K) in [(69632, 1536, 512), (69632, 512, 512), (69632, 512, 1024)]:
        A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
        Wp, ws = ops.pack_split_f16(W); Ap = ops.split_rows_f16(A)
        out = torch.empty(M, N, device=dev)
        run = lambda: ops.gemm_nt_split(Ap, Wp, ws, b, None, 0, out=out, a_packed=True)
        res = []
        st = (ctypes.c_ulonglong * 4)()
        ck = (ctypes.c_ulonglong * 2)()
        for rnd in range(2):
            for k in names:
                raw.pmce_gemm_ws_set_dbg(k)
                raw.pmce_gemm_ws_stats(st, 1); raw.pmce_gemm_ws_clk(ck, 1)
                t = timeit(run)
                raw.pmce_gemm_ws_stats(st, 1); raw.pmce_gemm_ws_clk(ck, 1)
                steps = 23.0 * 12 * (M + 191) // 192 * (N // 256) * (K // 16)  # compute-wave k-tiles of the 23 launches timed
                if rnd: res.append(f"{names[k]}: {t:.0f} [cw fail/step {st[0]/steps:.2f} polled {st[2]/steps:.2f} ld fail/step {st[1]/(steps/3):.2f}] clock {ck[0] / max(ck[1], 1) * 0.1:.2f} GHz")
        raw.pmce_gemm_ws_set_dbg(0)
        print(f"ablation {M} x {N} x {K} (us): " + "\n   ".join(res) + f" | timeouts {dlib.pmce_gemm_ws_timeouts()}", flush=True)
    lib.pmce_gemm_split_set_tuning(-1)

if "--timeline" in sys.argv:  # ablate build: shader clocks per wave role and phase (s_memtime instrumented, so slower than the product)
    import ctypes
    raw = ctypes.CDLL(diag.LIB_PATH)
    lib.pmce_gemm_split_set_tuning(3)
    for (M, N, K) in [(69632, 1536, 512), (69632, 512, 1024)]:
        A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
        Wp, ws = ops.pack_split_f16(W); Ap = ops.split_rows_f16(A)
        out = torch.empty(M, N, device=dev)
        run = lambda: ops.gemm_nt_split(Ap, Wp, ws, b, None, 0, out=out, a_packed=True)
        pr = (ctypes.c_ulonglong * 64)()
        for mode, name in ((128, "full"), (384, "62 of 64 stores per wave and tile"), (640, "56 of 64 stores"), (129, "no stores")):
            raw.pmce_gemm_ws_set_dbg(mode)
            run(); torch.cuda.synchronize(); raw.pmce_gemm_ws_prof(pr, 1)
            t = timeit(run, n=10); raw.pmce_gemm_ws_prof(pr, 1)
            launches = 13.0
            ktiles = ((M + 191) // 192) * (N // 256) * (K // 16) / 256.0  # k-tiles per workgroup (mean)
            print(f"timeline {M} x {N} x {K} [{name}] {t:.0f} us per launch; shader clocks per k-tile, mean over workgroups:")
            for w in range(16):
                v = [pr[4 * w + i] / launches / 256.0 for i in range(4)]
                if w < 12:
                    print(f"   compute wave {w:2d}: wait land {v[0]/ktiles:7.0f} | fragments {v[1]/ktiles:7.0f} | matrix issue {v[2]/ktiles:7.0f} | epilogue (per k-tile share) {v[3]/ktiles:7.0f}")
                else:
                    print(f"   loader wave  {w:2d}: wait stage {v[0]/ktiles:7.0f} | DMA issue {v[1]/ktiles:7.0f} | wait landed + post {v[2]/ktiles:7.0f}")
        raw.pmce_gemm_ws_set_dbg(0)
    lib.pmce_gemm_split_set_tuning(-1)

if "--check-opt" in sys.argv:  # ablate build: correctness of schedule variants (OPT bits) against the 4-wave kernel
    import ctypes
    raw = ctypes.CDLL(diag.LIB_PATH)
    for (M, N, K) in [(69632, 1536, 512), (40000, 640, 256)]:
        A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
        Wp, ws = ops.pack_split_f16(W); Ap = ops.split_rows_f16(A)
        run = lambda: ops.gemm_nt_split(Ap, Wp, ws, b, None, 0, a_packed=True)
        lib.pmce_gemm_split_set_tuning(0); ref = run().clone()
        lib.pmce_gemm_split_set_tuning(3)
        for o in (0, 1, 8, 9):
            raw.pmce_gemm_ws_set_dbg(200 + o)
            nb = []
            for _ in range(8):
                c = run()
                nb.append(int((c.view(torch.int32) != ref.view(torch.int32)).sum().item()))
            t = timeit(run)
            print(f"check {M} x {N} x {K} OPT={o}: mismatching elements per run {nb}  {t:.0f} us", flush=True)
        raw.pmce_gemm_ws_set_dbg(0)
    lib.pmce_gemm_split_set_tuning(-1)
