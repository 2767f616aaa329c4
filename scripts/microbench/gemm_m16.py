"""The split-f16 GEMM on v_mfma_f32_16x16x32_f16 (gemm_split_m16.hip) against the 32x32x16 kernel at the lifter's shapes: launch time,
maximum error of both against an fp64 product on sampled rows, and the largest difference between the two.
   python scripts/microbench/gemm_m16.py [C]"""
import sys
import os.path as osp
sys.path.insert(0, osp.dirname(osp.dirname(osp.dirname(osp.abspath(__file__)))))
import torch
from pmce_amd import _lib, ops
from scripts.microbench import diag      # the 16x16x32 kernel is in the diagnostics library (not in the product)

C = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda:0")
lib = _lib.load()
M = 256 * 16 * 17
shapes = [("qkv", 3 * C, C, 0, False, False), ("proj+res", C, C, 0, True, False), ("fc1 gelu packed", 2 * C, C, 1, False, True),
          ("fc2+res", C, 2 * C, 0, True, False), ("edge M", C, C, 0, True, False), ("proj, no res", C, C, 0, False, False),
          ("fc2, no res", C, 2 * C, 0, False, False), ("fc1 gelu fp32 out", 2 * C, C, 1, False, False)]


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for name, N, K, act, res, cpk in shapes:
    m = M if name != "edge M" else 5000 * 7 + 3
    g = torch.Generator().manual_seed(11)
    A = torch.randn(m, K, generator=g).to(dev)
    W = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    R = torch.randn(m, N, generator=g).to(dev) if res else None
    Wp, ws = ops.pack_split_f16(W)
    Ap = ops.split_rows_f16(A)
    outs, times, t128 = {}, {}, {}
    for mode in (0, 1):
        if mode == 0:
            run = lambda: ops.gemm_nt_split(Ap, Wp, ws, b, R, act, a_packed=True, c_packed=cpk)
            run128 = run
        else:
            run = lambda: diag.gemm_nt_split(1, Ap, Wp, ws, b, R, act, a_packed=True, c_packed=cpk, tile=0)
            run128 = lambda: diag.gemm_nt_split(1, Ap, Wp, ws, b, R, act, a_packed=True, c_packed=cpk, tile=1)
        outs[mode] = run()
        times[mode] = timeit(run)
        lib.pmce_gemm_split_set_tuning(1)        # the 128 x 128 block tile (64 x 64 wave tile: the 16x16x32 kernel defers its I3 there)
        t128[mode] = timeit(run128)
        lib.pmce_gemm_split_set_tuning(-1)
    rows = torch.randint(0, m, (512,), generator=g).to(dev)
    ref = A[rows].double() @ W.double().t() + b.double()
    if act:
        ref = torch.nn.functional.gelu(ref)
    if res:
        ref = ref + R[rows].double()
    val = {k: (ops.unsplit_rows_f16(v) if cpk else v.double()) for k, v in outs.items()}
    e0, e1 = ((val[k][rows] - ref).abs().max().item() for k in (0, 1))
    d = (val[0] - val[1]).abs().max().item()
    flops = 3 * 2.0 * m * N * K
    print(f"{name:16s} {m} x {N} x {K}: 32x32x16 {times[0]:7.1f} us ({flops / times[0] / 1e6:6.0f} TF)   16x16x32 {times[1]:7.1f} us ({flops / times[1] / 1e6:6.0f} TF)   "
          f"| 128x128 tile: {t128[0]:7.1f} / {t128[1]:7.1f} us | err vs fp64: {e0:.2e} / {e1:.2e}   max |difference| {d:.2e}   finite {bool(torch.isfinite(val[1]).all())}", flush=True)
