"""ctypes binding of scripts/microbench/libpmce_diag.so - the diagnostics library (pmce_amd.build.build_diag): the bystander / spinner
kernels of the matrix-pipe interference report and the f16-subnormal probe.  Not part of the product."""
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
