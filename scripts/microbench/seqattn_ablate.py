"""Ablation of lifter.hip's seq_attention kernel (timing only; variants compute garbage): what bounds it?"""
import ctypes as C, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
src = open(os.path.join(REPO, "pmce_amd/csrc/lifter.hip")).read()
def variant(v):
    s = src
    if v == "B":   # no key loop (staging + q load + store only)
        s = s.replace("  for (int j = 0; j < N; ++j) {\n    const float* kr", "  for (int j = 0; j < 0; ++j) {\n    const float* kr")
    if v == "C":   # key loop on ONE key (LDS/VALU work / N)
        s = s.replace("  for (int j = 0; j < N; ++j) {\n    const float* kr", "  for (int j = 0; j < 1; ++j) {\n    const float* kr")
    if v == "D":   # no K/V staging (LDS garbage), full compute
        s = s.replace("  for (int idx = tid; idx < N * C4; idx += 256) {", "  for (int idx = tid; idx < 0; idx += 256) {")
    if v == "E":   # no output store
        s = s.replace("    *reinterpret_cast<f32x4*>(dst + 4 * d4) = t;\n  }\n}", '    asm volatile("" :: "v"(t));\n  }\n}', 1)
    assert s != src or v == "A", v
    return s
libs = {}
for v in "ABCDE":
    d = f"/tmp/sa_abl_{v}"; os.makedirs(d, exist_ok=True)
    open(f"{d}/lifter.hip", "w").write(variant(v))
    for f in ("common.hpp", "common.cpp"):
        open(f"{d}/{f}", "w").write(open(os.path.join(REPO, "pmce_amd/csrc", f)).read())
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-x", "hip",
                        f"{d}/lifter.hip", f"{d}/common.cpp", "-o", f"{d}/lib.so"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lib = C.CDLL(f"{d}/lib.so")
    vp, i, l = C.c_void_p, C.c_int, C.c_longlong
    lib.pmce_seq_attention_f32.argtypes = [vp, vp, i, i, i, i, l, l, l, vp]
    libs[v] = lib
dev = torch.device("cuda:0")
desc = {"A": "full", "B": "no key loop", "C": "one key", "D": "no K/V staging", "E": "no store"}
B, T, J, Cc = 256, 16, 17, 256
qkvs = [torch.randn(B * T * J, 3 * Cc, device=dev) for _ in range(3)]; outs = [torch.empty(B * T * J, Cc, device=dev) for _ in range(3)]
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, args in (("spatial", (B * T, J, Cc, 0, J, 0, 1)), ("temporal", (B * J, T, Cc, J, 1, T * J, J))):
    best = {v: 1e9 for v in libs}
    for rnd in range(3):
        for v, lib in libs.items():
            call = lambda i: lib.pmce_seq_attention_f32(qkvs[i % 3].data_ptr(), outs[i % 3].data_ptr(), *args, st)
            assert call(0) == 0; torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(21): call(i)
            e1.record(); torch.cuda.synchronize()
            best[v] = min(best[v], e0.elapsed_time(e1) / 21)
    print(f"{name}: " + " | ".join(f"{v} {desc[v]}: {best[v]*1e3:6.1f}us" for v in libs), flush=True)
