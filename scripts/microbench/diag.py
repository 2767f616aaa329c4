"""ctypes binding of scripts/microbench/libpmce_diag.so - the diagnostics library (pmce_amd.build.build_diag): the two experimental
split-GEMM variants that are NOT in the product library (wave-specialised 192x256, 16x16x32 shape) and the bystander / spinner
kernels of the matrix-pipe interference report."""
import ctypes as C
import os.path as osp
import sys

HERE = osp.dirname(osp.abspath(__file__))
REPO = osp.dirname(osp.dirname(HERE))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
LIB_PATH = osp.join(HERE, "libpmce_diag.so")
_f, _i, _l, _s, _fl = C.c_void_p, C.c_int, C.c_longlong, C.c_void_p, C.c_float
PROTOTYPES = {
    "pmce_diag_gemm_nt_split_f16": [_i, _f, _f, _f, _f, _f, _f, _i, _i, _i, _l, _l, _i, _i, _i, _i, _s],
    "pmce_gemm_ws_timeouts": [],
    "pmce_dbg_victim": [_i, _f, _i, _i, _f, _s],
    "pmce_dbg_mfma_spin": [_i, _f, _i, _i, _s],
    "pmce_dbg_mfma_subnormal": [_fl, _fl, _f, _s],
    "pmce_last_error_string": [],
}
_lib = None


def load():
    global _lib
    if _lib is None:
        import torch  # noqa: F401  (one HIP runtime per process: torch's, see pmce_amd/_lib.py)
        if not osp.exists(LIB_PATH):
            from pmce_amd import build
            build.build_diag()
        lib = C.CDLL(LIB_PATH)
        for name, at in PROTOTYPES.items():
            fn = getattr(lib, name)
            fn.argtypes = at
            fn.restype = C.c_char_p if name == "pmce_last_error_string" else C.c_int
        _lib = lib
    return _lib


def check(rc, what="libpmce_diag"):
    if rc != 0:
        raise RuntimeError(f"{what} failed (rc={rc}): {(load().pmce_last_error_string() or b'').decode()}")


def gemm_nt_split(kind, A, Wp, wscale, bias, R, act=0, a_packed=True, c_packed=False, tile=0):
    """kind 0: wave-specialised kernel, kind 1: 16x16x32 shape.  Operands as pmce_amd.ops.gemm_nt_split."""
    import torch
    from pmce_amd import _lib as P
    M, K = A.shape
    N = Wp.shape[0]
    out = torch.empty(M, N, device=A.device, dtype=torch.float32)
    check(load().pmce_diag_gemm_nt_split_f16(kind, P.ptr(A), P.ptr(Wp), P.ptr(wscale), P.ptr(bias), P.ptr(R), P.ptr(out), M, N, K, K, N,
                                             act, int(a_packed), int(c_packed), tile, P.current_stream()))
    return out
