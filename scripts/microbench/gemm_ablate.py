"""Ablation of gemm_f32.hip (timing only, results are garbage by construction): variants are generated from the product
source by text substitution, built into throw-away .so files under /tmp and timed on the lifter shapes."""
import ctypes as C, os, subprocess, sys, re
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import torch
src = open(os.path.join(REPO, "pmce_amd/csrc/gemm_f32.hip")).read()
KEEP = 'asm volatile("" :: "v"(v));'
STORE2 = """        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v.x), rc, lane_off, EPI_BYTES(e, ldcb), 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v.y), rc, lane_off, EPI_BYTES(e + 1, ldcb), 0);"""
def variant(name):
    s = src
    if name >= "B":   # no epilogue stores (values kept alive)
        assert STORE2 in s
        s = s.replace(STORE2, '        asm volatile("" :: "v"(v.x), "v"(v.y));')
        s = s.replace("Cp[((r & 3) + 8 * (r >> 2)) * ldc] = v;", KEEP).replace("Cp[rr * ldc] = v;", KEEP)
    if name >= "C":   # no DMA inside the k-loop (LDS keeps the prologue's tile)
        s = s.replace("      gdma(kt + 1, buf ^ 1);\n", "      ;\n").replace("      set_ptrs(m_next, n_next);\n      gdma(0, buf ^ 1);\n", "      ;\n")
    if name >= "D":   # no wait / barrier in the k-loop
        s = s.replace("    dma_wait_and_sync();\n    buf ^= 1;\n", "    buf ^= 1;\n")
    if name >= "E":   # no ds_reads either: operands are whatever is in registers
        s = s.replace("for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(as + i * 32 * LD + co);",
                      "for (int i = 0; i < TM; ++i) a[i] = f32x4{as[0], as[1], as[2], as[3]};")
        s = s.replace("for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4*>(bs + j * 32 * LD + co);",
                      "for (int j = 0; j < TN; ++j) b[j] = f32x4{bs[0], bs[1], bs[2], bs[3]};")
    return s
libs = {}
for v in "ABCDE":
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
desc = {"G": "stores to a cache-resident region", "A": "baseline", "B": "- epilogue stores", "C": "- DMA", "D": "- wait+barrier", "E": "- ds_reads (MFMA only)"}
for name, M, N, K, act, res, tile in [("qkv 96x128", 69632, 768, 256, 0, False, 1), ("qkv 128x128", 69632, 768, 256, 0, False, 0), ("proj 64x64", 69632, 256, 256, 0, True, 3), ("fc2 64x64", 69632, 256, 512, 0, True, 3), ("gi0 128x128", 4096, 6144, 2048, 0, False, 0)]:
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
