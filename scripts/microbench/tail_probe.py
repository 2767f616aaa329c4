"""How much of a lifter product's launch is its last, partly filled round of tiles?  M = 69,632 rows (B = 256) are 544 row panels of 128:
on 512 persistent workgroups the N = 512 products have 2.125 tiles per workgroup (the busiest does 3), fc1 4.25, qkv 6.375.
Times: the whole product in one launch | the rows of the full rounds only | those + the remaining rows as a second launch (small tiles)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pmce_amd import _lib, ops

lib = _lib.load()
dev = torch.device("cuda:0")


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


M = 69632
for name, N, K, res, act, opack in (("proj", 512, 512, True, 0, False), ("fc2", 512, 1024, True, 0, False), ("fc1", 1024, 512, False, 1, True),
                                    ("qkv", 1536, 512, False, 0, False)):
    g = torch.Generator().manual_seed(3)
    A = ops.split_rows_f16(torch.randn(M, K, generator=g).to(dev))
    W = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    R = torch.randn(M, N, generator=g).to(dev) if res else None
    Wb, ws, _ = ops.pack_split_f16_blk(W)
    out = torch.empty(M, N, device=dev)
    tiles = (M // 128) * (N // 256)
    full = (tiles // 512) * 512 // (N // 256) * 128          # rows covered by the full rounds of 128 x 256 tiles

    def run(lo, hi, tile=-1):
        lib.pmce_gemm_split_set_tuning(tile)
        ops.gemm_nt_split_blk(A[lo:hi], Wb, ws, N, b, residual=R[lo:hi] if res else None, act=act, a_packed=True, c_packed=opack, out=out[lo:hi])
        lib.pmce_gemm_split_set_tuning(-1)

    t_all = timeit(lambda: run(0, M))
    t_full = timeit(lambda: run(0, full))
    t_tail = {t: timeit(lambda: run(full, M, t)) for t in (-1, 1, 2)}
    t_two = {t: timeit(lambda: (run(0, full), run(full, M, t))) for t in (-1, 1, 2)}
    ref = out.clone()
    run(0, M)
    same = torch.equal(ref, out)
    print(f"{name:5s} N={N:4d} K={K:4d}: {tiles} tiles = {tiles / 512:.3f} per workgroup; one launch {t_all:6.1f} us | rows < {full}: {t_full:6.1f} | "
          f"tail alone auto/128x128/64x128 {t_tail[-1]:5.1f}/{t_tail[1]:5.1f}/{t_tail[2]:5.1f} | two launches {t_two[-1]:6.1f}/{t_two[1]:6.1f}/{t_two[2]:6.1f} | bits equal {same}",
          flush=True)
