"""Packed-output (fc1) form of the split GEMM vs splitting the fp32-output form's result afterwards (debugging aid)."""
import sys, torch
sys.path.insert(0, ".")
from pmce_amd import ops, _lib
lib = _lib.load()
dev = "cuda"
torch.manual_seed(0)
M, N, K = 4352, 512, 256
A = torch.randn(M, K, device=dev)
W = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev)
Wp, ws = ops.pack_split_f16(W); Ap = ops.split_rows_f16(A)
lib.pmce_gemm_split_set_tuning(0)
f32 = ops.gemm_nt_split(Ap, Wp, ws, b, None, 1, a_packed=True)
lin = ops.gemm_nt_split(Ap, Wp, ws, b, None, 0, a_packed=True)
pk = ops.gemm_nt_split(Ap, Wp, ws, b, None, 1, a_packed=True, c_packed=True)
ref = ops.split_rows_f16(f32)
bad = pk.view(torch.int32) != ref.view(torch.int32)
print("differing dwords", int(bad.sum()), "of", bad.numel())
pl = pk.view(torch.float16).reshape(M, N // 16, 2, 16).float()
back = (pl[:, :, 0, :] + pl[:, :, 1, :] * 2.0 ** -11).reshape(M, N)
d = (back - f32)
print("value-level: max |packed value - fp32 value|", d.abs().max().item(), " elements differing", int((d != 0).sum()))
idx = (d != 0).nonzero()[:8].tolist()
g64 = torch.nn.functional.gelu(lin.double())
for r, c in idx:
    print(f"  ({r},{c}): pre-activation {lin[r,c].item():.9g}  fp32-output form {f32[r,c].item():.9g}  packed form {back[r,c].item():.9g}  fp64 gelu {g64[r,c].item():.12g}")
lib.pmce_gemm_split_set_tuning(-1)
