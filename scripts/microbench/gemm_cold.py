"""Is the isolated GEMM rate a warm-cache artefact?  Same product on ONE operand set vs rotating over several (GPU only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pmce_amd import ops
dev = torch.device("cuda:0")
for name, M, N, K, res in (("qkv", 69632, 768, 256, False), ("proj", 69632, 256, 256, True), ("fc2", 69632, 256, 512, True)):
    for nset in (1, 6):
        A = [torch.randn(M, K, device=dev) for _ in range(nset)]
        W = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev)
        R = [torch.randn(M, N, device=dev) for _ in range(nset)] if res else None
        out = [torch.empty(M, N, device=dev) for _ in range(nset)]
        best = 1e9
        for rnd in range(3):
            for i in range(nset): ops.gemm_nt(A[i], W, b, R[i] if res else None, 0, out=out[i])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 60
            e0.record()
            for i in range(n): ops.gemm_nt(A[i % nset], W, b, R[i % nset] if res else None, 0, out=out[i % nset])
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / n)
        print(f"{name} sets={nset}: {best*1e3:7.1f} us  {2.0*M*N*K/best/1e9:6.1f} TF", flush=True)
