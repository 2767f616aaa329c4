"""Timing-only ablation of the token-local decoder kernels' store / load patterns (variants compute garbage; GPU only)."""
import ctypes as C, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
src = open(os.path.join(REPO, "pmce_amd/csrc/coevo.hip")).read()
QKV_STORE = "        if (tile * 32 + tk < NV) *reinterpret_cast<f32x4*>(qkv + (tok0 + tk) * 192 + nt * 32 + 4 * (lane & 7)) = t;"
def variant(v):
    s = src
    if v == "B":   # adaln_qkv without its output stores
        assert QKV_STORE in s
        s = s.replace(QKV_STORE, '        asm volatile("" :: "v"(t));')
    if v == "C":   # adaln_qkv without the MFMAs
        s = s.replace("    tl_gemm<8, 6, LDW64>(sW, a, acc, n0, hb);", "    for (int nt = 0; nt < 6; ++nt) acc[nt][0] += a[nt];")
    return s
libs = {}
for v in "ABC":
    d = f"/tmp/coevo_abl_{v}"; os.makedirs(d, exist_ok=True)
    open(f"{d}/coevo.hip", "w").write(variant(v))
    for f in ("common.hpp", "common.cpp"):
        open(f"{d}/{f}", "w").write(open(os.path.join(REPO, "pmce_amd/csrc", f)).read())
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-x", "hip",
                        f"{d}/coevo.hip", f"{d}/common.cpp", "-o", f"{d}/lib.so"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lib = C.CDLL(f"{d}/lib.so")
    vp, i = C.c_void_p, C.c_int
    lib.pmce_adaln_qkv_f32.argtypes = [vp, vp, i, i, vp, vp, vp, i, vp]
    libs[v] = lib
dev = torch.device("cuda:0")
B, NV = 256, 431
xs = [torch.randn(B, NV, 64, device=dev) for _ in range(3)]
GB = torch.randn(B, 3072, device=dev); W = torch.randn(192, 64, device=dev) * 0.1; b = torch.randn(192, device=dev)
outs = [torch.empty(B, NV, 192, device=dev) for _ in range(3)]
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
desc = {"A": "full", "B": "no stores", "C": "no MFMA"}
best = {v: 1e9 for v in libs}
for rnd in range(3):
    for v, lib in libs.items():
        call = lambda i: lib.pmce_adaln_qkv_f32(xs[i % 3].data_ptr(), GB.data_ptr(), 3072, 4, W.data_ptr(), b.data_ptr(), outs[i % 3].data_ptr(), B, st)
        assert call(0) == 0; torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(21): call(i)
        e1.record(); torch.cuda.synchronize()
        best[v] = min(best[v], e0.elapsed_time(e1) / 21)
print("adaln_qkv: " + " | ".join(f"{v} {desc[v]}: {best[v]*1e3:6.1f}us" for v in libs), flush=True)
