import sys, os.path as osp
sys.path.insert(0, osp.dirname(osp.dirname(osp.dirname(osp.abspath(__file__)))))
import torch
from pmce_amd import ops
dev = torch.device("cuda:0")
C, J, B, Tn, H = 512, 17, 1, 16, 8
hd = C // H
M = B * Tn * J
g = torch.Generator().manual_seed(3)
qkv = (torch.randn(M, 3 * C, generator=g) * 1.7).to(dev)
planes = ops.split_rows_f16(qkv)
exact = ops.unsplit_rows_f16(planes)
x = exact.reshape(B, Tn, J, 3, H, hd)
q, k, v = (x[..., i, :, :].permute(0, 1, 3, 2, 4) for i in range(3))
a = ((q @ k.transpose(-2, -1)) * hd ** -0.5).softmax(-1) @ v
want = a.permute(0, 1, 3, 2, 4).reshape(M, C)
got_p = ops.seq_attention_split(planes, B * Tn, J, C, 0, J, 0, 1)
got = ops.unsplit_rows_f16(got_p)
err = (got - want).abs()
print("max err", err.max().item(), "mean", err.mean().item())
h16 = got_p.view(torch.float16).view(M, C // 16, 2, 16)
hi, lo = h16[:, :, 0].double().reshape(M, C), h16[:, :, 1].double().reshape(M, C) / 2048
whi = want.half().double()
print("hi == rne16(want):", (hi == whi).double().mean().item(), " hi == rtz?", ((hi - want).abs() < (want.abs() * 2 ** -10)).double().mean().item())
print("lo vs (want - hi): max", (lo - (want - hi)).abs().max().item(), " lo zero frac", (lo == 0).double().mean().item(), "lo abs mean", lo.abs().mean().item(), "want-hi abs mean", (want - hi).abs().mean().item())
bad = err > 2e-5
print("bad frac", bad.double().mean().item())
rows = bad.any(1).nonzero().flatten()
print("bad rows (token j):", sorted(set((rows % J).tolist())), "bad cols mod 64:", sorted(set((bad.any(0).nonzero().flatten() % 64).tolist()))[:64])
print("bad heads:", sorted(set((bad.any(0).nonzero().flatten() // 64).tolist())))
i = err.argmax().item(); r, c = divmod(i, C)
print("worst", r, c, "want", want[r, c].item(), "got", got[r, c].item(), "hi", hi[r, c].item(), "lo", lo[r, c].item())
