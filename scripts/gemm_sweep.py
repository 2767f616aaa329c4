"""Time pmce_gemm_nt_f32 on the path's GEMM shapes for every tile configuration (tuning aid, GPU only).
Configs are interleaved round-robin and the best of 4 rounds is reported, so clock ramp-up does not favour a column."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pmce_amd import _lib, ops

lib = _lib.load()

dev = torch.device("cuda:0")
shapes = [  # name, M, N, K, act, res
    ("qkv", 69632, 768, 256, 0, False), ("proj", 69632, 256, 256, 0, True), ("fc1", 69632, 512, 256, 1, False),
    ("fc2", 69632, 256, 512, 0, True), ("gi0", 4096, 6144, 2048, 0, False), ("gi1", 2304, 3072, 2048, 0, False),
    ("final", 256, 20670, 3360, 0, False), ("ada", 256, 3072, 2048, 0, False), ("imgfeat", 4096, 256, 2048, 0, False),
    ("qkv512", 69632, 1536, 512, 0, False), ("proj512", 69632, 512, 512, 0, True), ("fc1_512", 69632, 1024, 512, 1, False),
    ("fc2_512", 69632, 512, 1024, 0, True), ("qkvJ19", 77824, 768, 256, 0, False),
]
names = ["128x128", "96x128", "64x128", "64x64", "auto", "64x128/g3", "96x128/g3?", "64x64/g5"]
cfgs = [(0, None), (1, None), (2, None), (3, None), (None, None), (2, 3), (1, 3), (3, 5)]
only = sys.argv[1:] 
for name, M, N, K, act, res in shapes:
    if only and name not in only: continue
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev)
    R = torch.randn(M, N, device=dev) if res else None
    out = torch.empty(M, N, device=dev)
    best = [1e9] * len(cfgs)
    for rnd in range(4):
        for t, (tile, grid) in enumerate(cfgs):
            lib.pmce_gemm_set_tuning(-1 if tile is None else tile, 0 if grid is None else grid)
            ops.gemm_nt(A, W, b, R, act, out=out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            n = 10
            for _ in range(n): ops.gemm_nt(A, W, b, R, act, out=out)
            e1.record(); torch.cuda.synchronize()
            best[t] = min(best[t], e0.elapsed_time(e1) / n)
    lib.pmce_gemm_set_tuning(-1, 0)
    print(f"{name:8s} M={M} N={N} K={K}: " + " | ".join(f"{names[t]} {best[t]*1e3:7.1f}us {2.0*M*N*K/best[t]/1e9:6.1f}TF" for t in range(len(cfgs))), flush=True)
