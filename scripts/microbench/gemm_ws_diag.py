"""Packed-output (fc1) form of the wave-specialised kernel vs the 4-wave kernel vs splitting the fp32 result afterwards (debugging aid)."""
import sys, torch
sys.path.insert(0, ".")
from pmce_amd import ops, _lib
lib = _lib.load()
dev = "cuda"
torch.manual_seed(0)
M, N, K = 69632, 1024, 512
A = torch.randn(M, K, device=dev); A[::7] *= 1e-3
W = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
Wp, ws = ops.pack_split_f16(W); Ap = ops.split_rows_f16(A)
res = {}
for name, tile in (("4-wave", 0), ("ws", 3)):
    lib.pmce_gemm_split_set_tuning(tile)
    f32 = ops.gemm_nt_split(Ap, Wp, ws, b, None, 1, a_packed=True)
    pk = ops.gemm_nt_split(Ap, Wp, ws, b, None, 1, a_packed=True, c_packed=True)
    ref = ops.split_rows_f16(f32)
    res[name] = (f32.clone(), pk.clone())
    bad = pk.view(torch.int32) != ref.view(torch.int32)
    print(f"{name}: packed result vs split_rows(fp32 result): {int(bad.sum())} differing dwords")
    if bad.any():
        r, c = bad.nonzero()[0].tolist()
        h = pk.view(torch.float16)
        hr = ref.view(torch.float16)
        print("   first at row", r, "dword", c, "packed f16 pair", h[r, 2 * c:2 * c + 2].tolist(), "reference", hr[r, 2 * c:2 * c + 2].tolist())
        cols = bad.any(0).nonzero().flatten()
        print("   dword columns mod 16 histogram:", torch.bincount(cols % 16, minlength=16).tolist(), " rows mod 64 hist (first 16 bins of /4):", torch.bincount(bad.any(1).nonzero().flatten() % 64 // 4, minlength=16).tolist())
print("fp32 results equal between kernels:", torch.equal(res["4-wave"][0].view(torch.int32), res["ws"][0].view(torch.int32)))
print("packed results equal between kernels:", torch.equal(res["4-wave"][1].view(torch.int32), res["ws"][1].view(torch.int32)))
lib.pmce_gemm_split_set_tuning(-1)
