"""Time pmce_gemm_nt_f32 on the path's GEMM shapes for every tile configuration (tuning aid, GPU only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pmce_amd import ops, _lib

dev = torch.device("cuda:0")
shapes = [  # name, M, N, K, act, res
    ("qkv", 69632, 768, 256, 0, False), ("proj", 69632, 256, 256, 0, True), ("fc1", 69632, 512, 256, 1, False),
    ("fc2", 69632, 256, 512, 0, True), ("gi0", 4096, 6144, 2048, 0, False), ("gi1", 2304, 3072, 2048, 0, False),
    ("final", 256, 20670, 3360, 0, False), ("ada", 256, 3072, 2048, 0, False), ("imgfeat", 4096, 256, 2048, 0, False),
]
names = ["128x128", "96x128", "64x128", "64x64", "auto"]
for name, M, N, K, act, res in shapes:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev)
    R = torch.randn(M, N, device=dev) if res else None
    out = torch.empty(M, N, device=dev)
    row = []
    for t in range(5):
        if t < 4: os.environ["PMCE_GEMM_TILE"] = str(t)
        else: os.environ.pop("PMCE_GEMM_TILE", None)
        for _ in range(3): ops.gemm_nt(A, W, b, R, act, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 20
        for _ in range(n): ops.gemm_nt(A, W, b, R, act, out=out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        row.append(f"{names[t]} {ms*1e3:7.1f}us {2.0*M*N*K/ms/1e9:6.1f}TF")
    print(f"{name:8s} M={M} N={N} K={K}: " + " | ".join(row), flush=True)
