"""Stress: one small-M split GEMM many times, bitwise against the first run; with / without a concurrent stream."""
import sys, torch
sys.path.insert(0, ".")
from pmce_amd import ops, _lib
lib = _lib.load()
dev = "cuda"
torch.manual_seed(0)
for (M, N, K) in [(64, 3072, 2048), (64, 20670, 3360), (1024, 6144, 2048), (576, 3072, 2048)]:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev)
    Wp, ws = ops.pack_split_f16(W)
    for tile in (2, 1, 0):
        lib.pmce_gemm_split_set_tuning(tile)
        for skew in (-1, 0):
            lib.pmce_gemm_split_set_skew(skew)
            ref = ops.gemm_nt_split(A, Wp, ws, b).clone()
            bad = 0; worst = 0.0
            for r in range(200):
                out = ops.gemm_nt_split(A, Wp, ws, b)
                d = (out - ref).abs().max().item()
                bad += d != 0; worst = max(worst, d)
            e64 = (ref.double() - (A.double() @ W.double().t() + b.double())).abs().max().item()
            print(f"{M}x{N}x{K} tile{tile} skew{skew}: {bad}/200 differ (worst {worst:.2e}); ref vs fp64 {e64:.2e}", flush=True)
lib.pmce_gemm_split_set_tuning(-1); lib.pmce_gemm_split_set_skew(-1)
