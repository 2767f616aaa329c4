"""Ablation of gru.hip's step kernel (timing only; the variants compute garbage): where do the cycles of one GRU step go?"""
import ctypes as C, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
src = open(os.path.join(REPO, "pmce_amd/csrc/gru.hip")).read()
WAITS = ['if (NQ == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");', 'else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");',
         'asm volatile("s_waitcnt vmcnt(0)" ::: "memory");']
def variant(v):
    s = src
    nq = v[1]
    s = s.replace('getenv("PMCE_GRU_NQ") ? atoi(getenv("PMCE_GRU_NQ")) : 4', nq)
    if v[0] == "B":   # DMA issued but never waited for
        for w in WAITS: s = s.replace(w, ";")
    if v[0] == "C":   # no DMA at all: ds_read + MFMA only
        s = s.replace("        dma_stage(kt + 1);\n", "").replace("    dma_stage(0);\n", "")
        for w in WAITS: s = s.replace(w, ";")
    if v[0] == "D":   # DMA + waits only: no ds_read / MFMA
        s = s.replace("for (int g8 = 0; g8 < KS / 8; ++g8) {", "for (int g8 = 0; g8 < 0; ++g8) {")
    return s
libs = {}
for v in ("A2", "A4", "C2", "C4", "D4"):
    d = f"/tmp/gru_abl_{v}"; os.makedirs(d, exist_ok=True)
    open(f"{d}/gru.hip", "w").write(variant(v))
    for f in ("common.hpp", "common.cpp"):
        open(f"{d}/{f}", "w").write(open(os.path.join(REPO, "pmce_amd/csrc", f)).read())
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-x", "hip",
                        f"{d}/gru.hip", f"{d}/common.cpp", "-o", f"{d}/lib.so"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lib = C.CDLL(f"{d}/lib.so")
    vp, i, l = C.c_void_p, C.c_int, C.c_longlong
    lib.pmce_gru_step_f32.argtypes = [vp] * 10 + [l, l, i, i, i, vp]
    libs[v] = lib
dev = torch.device("cuda:0")
desc = {"A": "full", "B": "no vmcnt waits", "C": "no DMA (ds_read+MFMA)", "D": "DMA+waits only"}
desc = {v: desc[v[0]] + " NQ=" + v[1] for v in libs}
for B in (256, 1024):
    H = 1024
    gi = [torch.randn(B, 3 * H, device=dev) for _ in range(2)]
    whh = [torch.randn(3 * H, H, device=dev) * 0.03 for _ in range(2)]
    bhh = [torch.randn(3 * H, device=dev) for _ in range(2)]
    hp = [torch.randn(B, H, device=dev) for _ in range(2)]
    ho = [torch.empty(B, H, device=dev) for _ in range(2)]
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    best = {v: 1e9 for v in libs}
    for rnd in range(3):
        for v, lib in libs.items():
            call = lambda: lib.pmce_gru_step_f32(gi[0].data_ptr(), gi[1].data_ptr(), whh[0].data_ptr(), whh[1].data_ptr(), bhh[0].data_ptr(),
                                                 bhh[1].data_ptr(), hp[0].data_ptr(), hp[1].data_ptr(), ho[0].data_ptr(), ho[1].data_ptr(),
                                                 3 * H, H, B, H, 2, st)
            call(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): call()
            e1.record(); torch.cuda.synchronize()
            best[v] = min(best[v], e0.elapsed_time(e1) / 20)
    fl = 2.0 * B * 3 * H * H * 2
    print(f"B={B}: " + " | ".join(f"{v} {desc[v]}: {best[v]*1e3:6.1f}us {fl/best[v]/1e9:6.1f}TF" for v in libs), flush=True)
