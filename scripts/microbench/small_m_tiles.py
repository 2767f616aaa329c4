"""Launch time of the small-M products of a forward (final product, AdaLN parameters, layer-1 GRU projections, imgfeat_embed) under each
forced tile shape of the split GEMM against the automatic choice."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pmce_amd import _lib, ops

lib = _lib.load()
dev = torch.device("cuda:0")


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


shapes = [("gi1 B=256", 2304, 3072, 2048), ("gi1 B=64", 576, 3072, 2048), ("ada B=256", 256, 3072, 2048), ("final B=256", 256, 20670, 3360),
          ("final B=64", 64, 20670, 3360), ("ie B=256 (fp32 A)", 4096, 512, 2048), ("qkv B=1", 272, 1536, 512), ("qkv B=8", 2176, 1536, 512),
          ("fc2 B=8", 2176, 512, 1024), ("proj B=1", 272, 512, 512), ("fc1 B=1", 272, 2048, 512), ("K=1024 B=1", 272, 512, 1024), ("fc2 B=1", 272, 512, 2048),
          ("K=4096 B=1", 272, 512, 4096)]
if len(sys.argv) > 1:   # only the shapes whose name contains the argument, e.g. "B=1"
    shapes = [s for s in shapes if sys.argv[1] in s[0]]
for name, M, N, K in shapes:
    g = torch.Generator().manual_seed(3)
    A = torch.randn(M, K, generator=g).to(dev)
    W = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    Wb, ws, _ = ops.pack_split_f16_blk(W)
    out = torch.empty(M, N, device=dev)
    res = {}
    for tile in (-1, 0, 1, 2):
        lib.pmce_gemm_split_set_tuning(tile)
        res[tile] = timeit(lambda: ops.gemm_nt_split_blk(A, Wb, ws, N, b, out=out))
    lib.pmce_gemm_split_set_tuning(-1)
    print(f"{name:20s} {M:5d} x {N:5d} x {K:4d}: auto {res[-1]:7.1f} us | 128x256 {res[0]:7.1f} | 128x128 {res[1]:7.1f} | 64x128 {res[2]:7.1f}", flush=True)
