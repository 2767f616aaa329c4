"""A/B of two builds of pmce_gemm_nt_f32 (GPU only): interleaved timing on the path's shapes plus a max-abs comparison of
their outputs.  usage: gemm_ab.py name=path.so name=path.so ... [-- shape ...]"""
import ctypes as C, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

args = sys.argv[1:]
only = []
if "--" in args:
    only = args[args.index("--") + 1:]
    args = args[:args.index("--")]
libs = {}
for a in args:
    n, p = a.split("=")
    lib = C.CDLL(os.path.join(REPO, p) if not os.path.isabs(p) else p)
    vp, i, l = C.c_void_p, C.c_int, C.c_longlong
    lib.pmce_gemm_nt_f32.argtypes = [vp] * 5 + [i, i, i, l, i, l, i, i, l, l, i, l, l, i, l, l, l, l, vp]
    libs[n] = lib
dev = torch.device("cuda:0")
shapes = [  # name, M, N, K, act, res, forced tile (-1 auto)
    ("qkv", 69632, 768, 256, 0, False, -1), ("qkv/128", 69632, 768, 256, 0, False, 0), ("qkv/96", 69632, 768, 256, 0, False, 1),
    ("qkv/64", 69632, 768, 256, 0, False, 2), ("proj", 69632, 256, 256, 0, True, -1), ("fc1", 69632, 512, 256, 1, False, -1),
    ("fc2", 69632, 256, 512, 0, True, -1), ("gi0", 4096, 6144, 2048, 0, False, -1), ("gi1", 2304, 3072, 2048, 0, False, -1),
    ("final", 256, 20670, 3360, 0, False, -1), ("ada", 256, 3072, 2048, 0, False, -1), ("imgfeat", 4096, 256, 2048, 0, False, -1),
    ("qkv512", 69632, 1536, 512, 0, False, -1), ("ragged", 1000, 333, 96, 1, True, -1),
    ("qkv/64g3", 69632, 768, 256, 0, False, 2 + 30), ("fc1/64g3", 69632, 512, 256, 1, False, 2 + 30),  # +10*g: PMCE_GEMM_GRID=g
]
for name, M, N, K, act, res, tile in shapes:
    if only and name not in only:
        continue
    os.environ.pop("PMCE_GEMM_GRID", None)
    if tile >= 10:
        os.environ["PMCE_GEMM_GRID"] = str(tile // 10)
        tile %= 10
    if tile >= 0: os.environ["PMCE_GEMM_TILE"] = str(tile)
    else: os.environ.pop("PMCE_GEMM_TILE", None)
    torch.manual_seed(0)
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev)
    R = torch.randn(M, N, device=dev) if res else None
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    outs = {v: torch.zeros(M, N, device=dev) for v in libs}
    def call(v):
        rc = libs[v].pmce_gemm_nt_f32(A.data_ptr(), W.data_ptr(), b.data_ptr(), R.data_ptr() if res else None, outs[v].data_ptr(),
                                      M, N, K, K, K, N, act, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, st)
        assert rc == 0
    best = {v: 1e9 for v in libs}
    for rnd in range(4):
        for v in libs:
            call(v); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): call(v)
            e1.record(); torch.cuda.synchronize()
            best[v] = min(best[v], e0.elapsed_time(e1) / 10)
    ref = (A.double() @ W.double().T + b.double())
    if act: ref = torch.nn.functional.gelu(ref)
    if res: ref = ref + R.double()
    errs = {v: float((outs[v].double() - ref).abs().max()) for v in libs}
    print(f"{name:8s} " + " | ".join(f"{v}: {best[v]*1e3:7.1f}us {2.0*M*N*K/best[v]/1e9:6.1f}TF err {errs[v]:.1e}" for v in libs), flush=True)
