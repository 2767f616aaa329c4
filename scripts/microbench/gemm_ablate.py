"""Ablation of gemm_f32.hip (timing only, results are garbage by construction): variants are generated from the product
source by text substitution, built into throw-away .so files under /tmp and timed on the lifter shapes."""
import ctypes as C, os, subprocess, sys, re
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import torch
src = open(os.path.join(REPO, "pmce_amd/csrc/gemm_f32.hip")).read()
KEEP = 'asm volatile("" :: "v"(v));'
def variant(name):
    s = src
    if name == "G":   # same number of store instructions, but into a small per-workgroup region (cache-resident): issue vs HBM
        s = s.replace("pend.c = C + o;", "pend.c = C + (long long)blockIdx.x * 16384 + (o & 4095);")
        s = s.replace("pend.c[EPI_OFF(e)] = v;", "pend.c[(EPI_OFF(e)) & 8191] = v;")
        return s
    if name >= "B":   # no epilogue stores (values kept alive)
        s = s.replace("pend.c[EPI_OFF(e)] = v;", KEEP)
        s = s.replace("Cp[((r & 3) + 8 * (r >> 2)) * ldc] = v;", KEEP).replace("Cp[rr * ldc] = v;", KEEP)
    if name >= "C":   # no global loads inside the k-loop (registers keep the prologue's tile)
        s = s.replace("      gload(kt + 1);\n", "      ;\n").replace("      set_ptrs(m_next, n_next);\n      gload(0);\n", "      ;\n")
    if name >= "D":   # no LDS writes
        s = s.replace("    if (loaded) lstore(buf ^ 1);\n", "")
    if name >= "E":   # no barrier in the k-loop
        s = s.replace("    if (loaded) lstore(buf ^ 1);\n", "").replace("    __syncthreads();\n    buf ^= 1;\n", "    buf ^= 1;\n")
    if name >= "F":   # no ds_reads either: operands are whatever is in registers
        s = re.sub(r"for \(int i = 0; i < TM; \+\+i\) a\[i\] = \*reinterpret_cast<const f32x4\*>\(as \+ i \* 32 \* LD \+ 8 \* g\);",
                   "for (int i = 0; i < TM; ++i) a[i] = f32x4{as[0], as[1], as[2], as[3]};", s)
        s = re.sub(r"for \(int j = 0; j < TN; \+\+j\) b\[j\] = \*reinterpret_cast<const f32x4\*>\(bs \+ j \* 32 \* LD \+ 8 \* g\);",
                   "for (int j = 0; j < TN; ++j) b[j] = f32x4{bs[0], bs[1], bs[2], bs[3]};", s)
    return s
libs = {}
for v in "ABG":
    d = f"/tmp/gemm_abl_{v}"; os.makedirs(d, exist_ok=True)
    open(f"{d}/gemm_f32.hip", "w").write(variant(v))
    for f in ("common.hpp", "common.cpp"):
        open(f"{d}/{f}", "w").write(open(os.path.join(REPO, "pmce_amd/csrc", f)).read())
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-x", "hip",
                        f"{d}/gemm_f32.hip", f"{d}/common.cpp", "-o", f"{d}/lib.so"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lib = C.CDLL(f"{d}/lib.so")
    vp, i, l = C.c_void_p, C.c_int, C.c_longlong
    lib.pmce_gemm_nt_f32.argtypes = [vp] * 5 + [i, i, i, l, i, l, i, i, l, l, i, l, l, i, l, l, l, l, vp]
    libs[v] = lib
dev = torch.device("cuda:0")
desc = {"G": "stores to a cache-resident region", "A": "baseline", "B": "- epilogue stores", "C": "- global loads", "D": "- LDS writes", "E": "- barrier", "F": "- ds_reads (MFMA only)"}
for name, M, N, K, act, res, tile in [("qkv 96x128", 69632, 768, 256, 0, False, 1), ("qkv 128x128", 69632, 768, 256, 0, False, 0), ("qkv 64x128", 69632, 768, 256, 0, False, 2), ("fc1 96x128", 69632, 512, 256, 1, False, 1)]:
    os.environ["PMCE_GEMM_TILE"] = str(tile)
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev)
    R = torch.randn(M, N, device=dev) if res else None
    out = torch.empty(M, N, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    row = []
    best = {v: 1e9 for v in libs}
    for rnd in range(3):
        for v, lib in libs.items():
            call = lambda: lib.pmce_gemm_nt_f32(A.data_ptr(), W.data_ptr(), b.data_ptr(), R.data_ptr() if res else None, out.data_ptr(), M, N, K, K, K, N, act, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, st)
            call(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): call()
            e1.record(); torch.cuda.synchronize()
            best[v] = min(best[v], e0.elapsed_time(e1) / 10)
    print(f"{name:12s} " + " | ".join(f"{v} {desc[v]}: {best[v]*1e3:6.1f}us {2.0*M*N*K/best[v]/1e9:6.1f}TF" for v in libs), flush=True)
