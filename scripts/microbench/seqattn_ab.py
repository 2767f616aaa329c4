"""seq_attention at the path's full sizes (B = 256 clips): time per launch and effective HBM rate, spatial and temporal
form, C = 256 / 512, J = 17 / 19.  A/B: run once as is and once with PMCE_SEQ_ATTN_V1=1 (the one-query-per-lane kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pmce_amd import ops

dev = torch.device("cuda:0")
B, T = 256, 16
tag = "v1 (one query per lane)" if os.environ.get("PMCE_SEQ_ATTN_V1") else "v2 (query pair per lane)"
for C, J in ((256, 17), (512, 17), (256, 19)):
    M = B * T * J
    qkv = torch.randn(M, 3 * C, device=dev)
    forms = {"spatial": (B * T, J, C, 0, J, 0, 1), "temporal": (B * J, T, C, J, 1, T * J, J)}
    for name, a in forms.items():
        for _ in range(3):
            out = ops.seq_attention(qkv, *a)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            out = ops.seq_attention(qkv, *a)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        byt = M * 4 * C * 4
        print(f"{tag}: C={C} J={J} {name:8s} {us:7.1f} us   {byt / us / 1e6:6.2f} TB/s (q|k|v read + out written = {byt / 1e6:.0f} MB)")
