"""Build libpmce_hip.so (hand-written HIP for gfx950) in-tree with hipcc.  No torch extension machinery: the
library is a plain C-ABI shared object loaded with ctypes (pmce_amd/_lib.py)."""
from __future__ import annotations

import os
import os.path as osp
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = osp.dirname(osp.abspath(__file__))
CSRC = osp.join(HERE, "csrc")
LIB = osp.join(HERE, "libpmce_hip.so")
SOURCES = ["common.cpp", "gemm_f32.hip", "gemm_split_f16.hip", "gemm_split_small.hip", "seq_attention_mfma.hip", "lifter.hip", "gru.hip", "coevo.hip", "metrics.hip", "model.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# On MI355X waves that execute f16 matrix instructions disturb packed-fp32 (v_pk_*_f32) arithmetic of OTHER waves on the same CU
#  - waves of other kernels and waves of the same kernel that are in a vector phase while their neighbours are
# in the matrix phase.  No packed-fp32 instruction is generated for ANY kernel of the library, which is what allows its kernels to
# overlap each other (two streams inside a forward, pipeline lanes); tests/test_host_logic.py checks the device code.
NO_PACKED_FP32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
# PMCE_PACKED_FP32_FILES="coevo.hip,..." (A/B builds only, scripts/build_ab.sh): compile those files WITH compiler-chosen packed fp32
_PACKED_AB = {f for f in os.environ.get("PMCE_PACKED_FP32_FILES", "").split(",") if f}
FILE_FLAGS = {src: ([] if src in _PACKED_AB else NO_PACKED_FP32) for src in SOURCES if src.endswith(".hip")}

# The diagnostics library (NOT the product): the bystander kernels of the matrix-pipe interference report, which ARE packed-fp32 code on
# purpose, and the f16-subnormal probe.  (The two experimental split-GEMM variants it carried until round 4 - wave-specialised 192x256,
# 16x16x32 - are in the history: measured, not faster, profiles/r03_a_gemm_ws_*, r02_k_*.)
DIAG_DIR = osp.join(HERE, "..", "scripts", "microbench")
DIAG_LIB = osp.join(DIAG_DIR, "libpmce_diag.so")
DIAG_SOURCES = ["dbg_victims.hip"]
DIAG_FILE_FLAGS = {src: NO_PACKED_FP32 for src in DIAG_SOURCES if src != "dbg_victims.hip"}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (osp.isabs(c) and osp.exists(c) or not osp.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def source_id() -> str:
    """sha256 over the library's sources (csrc/*.hip|cpp|hpp + include/pmce_hip.h + the flags), first 16 hex digits: compiled into the
    library (pmce_build_id()) and recorded by the profiler summaries under profiles/, so that counters can be matched to a build."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".hpp")))
    for f in files + [osp.join("..", "..", "include", "pmce_hip.h")]:
        h.update(f.encode())
        with open(osp.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS + NO_PACKED_FP32 + os.environ.get("PMCE_EXTRA_HIPCC_FLAGS", "").split() + sorted(_PACKED_AB)).encode())
    return h.hexdigest()[:16]


def needs_build() -> bool:
    if not osp.exists(LIB):
        return True
    t = osp.getmtime(LIB)
    deps = [osp.join(CSRC, f) for f in os.listdir(CSRC)] + [osp.join(HERE, "..", "include", "pmce_hip.h")]
    return any(osp.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = _hipcc()
    objdir = osp.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    sid = source_id()

    def compile_one(src):
        obj = osp.join(objdir, osp.splitext(src)[0] + ".o")
        extra = os.environ.get("PMCE_EXTRA_HIPCC_FLAGS", "").split()
        if src == "common.cpp":
            extra = extra + [f'-DPMCE_BUILD_ID="{sid}"']
        cmd = [hipcc, *FLAGS, *FILE_FLAGS.get(src, []), *extra, "-x", "hip", "-c", osp.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        # the host pass of a -target-feature meant for the device prints "not a recognized feature for this target (ignoring)"
        err = "".join(l for l in r.stderr.splitlines(True) if "packed-fp32-ops' is not a recognized feature" not in l)
        if verbose and err.strip():
            sys.stderr.write(err)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


def build_diag(force: bool = False, verbose: bool = True) -> str:
    """scripts/microbench/libpmce_diag.so: diagnostics only (the bystander report of the GPU suite loads it); links its own copy
    of common.cpp (error string, launch check), shares the product's headers."""
    srcdir = osp.join(DIAG_DIR, "csrc")
    deps = [osp.join(srcdir, f) for f in DIAG_SOURCES] + [osp.join(CSRC, f) for f in ("common.cpp", "common.hpp", "gemm_split_common.hpp")]
    if not force and osp.exists(DIAG_LIB) and all(osp.getmtime(d) <= osp.getmtime(DIAG_LIB) for d in deps):
        return DIAG_LIB
    hipcc = _hipcc()
    objdir = osp.join(srcdir, "build")
    os.makedirs(objdir, exist_ok=True)

    def compile_one(item):
        d, src = item
        obj = osp.join(objdir, osp.splitext(src)[0] + ".o")
        extra = os.environ.get("PMCE_EXTRA_HIPCC_FLAGS", "").split()
        cmd = [hipcc, *FLAGS, "-I", CSRC, *DIAG_FILE_FLAGS.get(src, []), *extra, "-x", "hip", "-c", osp.join(d, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        err = "".join(l for l in r.stderr.splitlines(True) if "packed-fp32-ops' is not a recognized feature" not in l)
        if verbose and err.strip():
            sys.stderr.write(err)
        return obj

    with ThreadPoolExecutor(max_workers=5) as ex:
        objs = list(ex.map(compile_one, [(srcdir, s) for s in DIAG_SOURCES] + [(CSRC, "common.cpp")]))
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic", *objs, "-o", DIAG_LIB], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return DIAG_LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
    print(build_diag(force="--force" in sys.argv))
