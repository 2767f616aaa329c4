"""Does a long back-to-back run of one product hold the short-run rate?  (clock/power check; GPU only)"""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pmce_amd import ops
dev = torch.device("cuda:0")
M, N, K = 69632, 768, 256
A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev)
out = torch.empty(M, N, device=dev)
def smi():
    try:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout
        return " ; ".join(l.strip() for l in r.splitlines() if "sclk" in l or "Power" in l or "mclk" in l)
    except Exception as e:
        return repr(e)
for n in (10, 100, 1000, 4000):
    ops.gemm_nt(A, W, b, None, 0, out=out); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        ops.gemm_nt(A, W, b, None, 0, out=out)
        if i == n // 2 and n >= 1000: print("   mid-run:", smi(), flush=True)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / n
    print(f"n={n:5d}: {t*1e3:7.1f} us/launch  {2.0*M*N*K/t/1e9:6.1f} TF", flush=True)
print("idle:", smi())
