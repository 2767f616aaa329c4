"""Is the isolated GEMM rate a warm-cache artefact?  Same product on ONE operand set vs rotating over several (GPU only).
TILES=0,1,2,3 forces the tile configurations in turn (default: the launcher's own choice)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pmce_amd import ops

dev = torch.device("cuda:0")
tiles = [int(t) for t in os.environ.get("TILES", "-1").split(",")]
shapes = (("qkv", 69632, 768, 256, 0, False), ("proj", 69632, 256, 256, 0, True), ("fc1", 69632, 512, 256, 1, False),
          ("fc2", 69632, 256, 512, 0, True), ("qkv512", 69632, 1536, 512, 0, False), ("proj512", 69632, 512, 512, 0, True),
          ("fc1_512", 69632, 1024, 512, 1, False), ("fc2_512", 69632, 512, 1024, 0, True))
only = [a for a in sys.argv[1:]]
shapes = [sh for sh in shapes if not only or sh[0] in only]
for name, M, N, K, act, res in shapes:
    for tile in tiles:
        from pmce_amd import _lib
        _lib.load().pmce_gemm_set_tuning(tile, 0)
        row = []
        for nset in (1, 6):
            A = [torch.randn(M, K, device=dev) for _ in range(nset)]
            W = torch.randn(N, K, device=dev) * K ** -0.5
            b = torch.randn(N, device=dev)
            R = [torch.randn(M, N, device=dev) for _ in range(nset)] if res else None
            out = [torch.empty(M, N, device=dev) for _ in range(nset)]
            best = 1e9
            for rnd in range(3):
                for i in range(nset):
                    ops.gemm_nt(A[i], W, b, R[i] if res else None, act, out=out[i])
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                n = 60
                e0.record()
                for i in range(n):
                    ops.gemm_nt(A[i % nset], W, b, R[i % nset] if res else None, act, out=out[i % nset])
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / n)
            row.append(f"sets={nset}: {best*1e3:7.1f} us {2.0*M*N*K/best/1e9:6.1f} TF")
        print(f"{name:5s} tile {tile:2d}  " + " | ".join(row), flush=True)
