#!/usr/bin/env python3
"""bench.py — PMCE hot-path throughput on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU over RCCL.  Either the caller launches the ranks (`python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N`, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* in the environment) or, when WORLD_SIZE is
not set, this script spawns its own N ranks on 127.0.0.1 and relays rank 0's JSON line.

A "step" is one pass of the full two-stream hot path (temporal pose encoder + CoEvoDecoder + 6890-vertex upsample +
J_regressor projection) over one batch of synthetic 16-frame clips per GPU, inputs resident in HBM.  The headline
configuration is the one BASELINE.json's north_star quotes throughput on — (B=256, T=16, J=17, C=512), BASELINE configs[2]
batch — and the same line carries a second complete record for C=256, the width every reference config ships
(`/root/reference/lib/core/config.py:59`).  Rank 0 prints ONE JSON line: whole-job clips/s (median of `--windows` timed
windows of exactly K steps each, every window bracketed by barrier + synchronize, MAX over ranks), the roofline of the
dominant kernel class and of the north-star cross-attention kernel (HIP events recorded inside this run on the launch
stream), small-batch latency, and the oracle's CPU baseline on the host cores (rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import socket
import statistics
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_F32_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: fp32 MFMA == fp32 vector peak
PEAK_F16_TFLOPS = 2500.0    # same guide: dense f16/bf16 MFMA peak (v_mfma_f32_32x32x16_f16)
PEAK_HBM_GBS = 8000.0       # HBM3E peak (same guide; ~6.3 TB/s is what a streaming copy reaches)
CLOCK_GHZ = 2.4
# what v_mfma_f32_32x32x16_f16 sustains from REGISTER operands holding random f16 bit patterns, pipe 98-100 % occupied, at the
# 1.42-1.54 GHz the chip then holds (scripts/microbench/mfma_peak.hip, profiles/r03_d_mfma_peak_register_operands.txt; zeros: 2,357)
MEASURED_F16_CEILING_TFLOPS = 1582.0
N_SIMD = 1024               # 256 CUs x 4 SIMDs; one v_mfma_f32_32x32x2_f32 occupies a SIMD's matrix pipe for 64 cycles


# ------------------------------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` without a launcher -> N ranks, one per GPU
# ------------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n: int, argv) -> int:
    """Spawn n copies of this script (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set, rendezvous on 127.0.0.1), relay rank 0's
    stdout, return the worst exit code.  A rank that dies takes the others down (no orphan holding a GPU)."""
    import tempfile
    port = _free_port()
    out0 = tempfile.TemporaryFile(mode="w+")
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), PMCE_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [c for c in codes if c not in (None, 0)]
            if bad:
                rc = bad[0]
                break
            if all(c == 0 for c in codes):
                break
            time.sleep(0.1)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            try:
                p.wait(timeout=30)
            except Exception:  # noqa: BLE001
                pass
    out0.seek(0)
    sys.stdout.write(out0.read())
    sys.stdout.flush()
    return rc


def bind_rank_to_cpus(local: int, nlocal: int):
    """One process per GPU: give rank `local` of `nlocal` its own contiguous slice of the CPUs this process may run on, so that
    the ranks' launch threads (and rank 0's CPU baseline) do not migrate over each other.  Returns the slice (or None when the
    platform has no sched_setaffinity / there is one rank)."""
    if nlocal <= 1 or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        cpus = sorted(os.sched_getaffinity(0))
        per = max(1, len(cpus) // nlocal)
        mine = cpus[local * per:(local + 1) * per] or cpus
        os.sched_setaffinity(0, mine)
        return mine
    except OSError:
        return None


def shared_state_dict(J, C, local, barrier, nonce=""):
    """The synthetic weights (412 MB at C = 256, 6-7 s of host time to generate) made ONCE per node: local rank 0 writes them to
    /dev/shm, the other ranks map the file (torch.load(mmap=True)) instead of regenerating them 8 times.  `barrier` is a callable all
    ranks of the node call; the file is removed by its writer after the second barrier."""
    import torch
    from pmce_amd import synth
    tag = os.environ.get("MASTER_PORT", "0")
    path = f"/dev/shm/pmce_bench_weights_J{J}_C{C}_{os.getuid()}_{tag}_{nonce}.pt"     # nonce: per run (rank 0's, broadcast): a file a crashed
    sd = None                                                                             # earlier run left behind is never picked up
    if local == 0:
        sd = synth.make_state_dict(synth.pmce_spec(J, C, 3), seed=123)
        try:
            if os.path.exists(path):
                os.unlink(path)
            torch.save(sd, path + ".tmp")
            os.replace(path + ".tmp", path)
        except OSError:
            path = None
    barrier()
    if local != 0:
        try:
            sd = torch.load(path, mmap=True, weights_only=True)
        except Exception:  # noqa: BLE001  (no /dev/shm, or the writer failed): generate locally
            sd = synth.make_state_dict(synth.pmce_spec(J, C, 3), seed=123)
    return sd, path


# ------------------------------------------------------------------------------------------------------------------
# analytical work (what measured times are divided into)
# ------------------------------------------------------------------------------------------------------------------
def gemm_lifter_flops(B, J, C, depth=3, T=16, F=2048, streaming=False):
    """2MNK of the lifter's products per step.  streaming: a WINDOW batch of the stride-1 mode - imgfeat_embed and SpatialBlocks[0]
    were computed once per frame (pmce_stream_precompute), the window pass runs the other 2*depth - 1 blocks."""
    M = B * T * J
    block = 2.0 * M * C * 3 * C + 2.0 * M * C * C + 2 * 2.0 * M * C * 2 * C
    if streaming:
        return (depth * 2 - 1) * block
    return 2.0 * B * T * F * C + depth * 2 * block


# kernel function behind each timing class of model.cpp: the four GEMM classes are one kernel - gemm_split_kernel in the
# split-f16 mode (three f16 matrix products per fp32 product), gemm_nt_kernel on the fp32 matrix pipe
GEMM_CLASSES = ("gemm_lifter", "gemm_gru_in", "gemm_ada", "gemm_final")


def kernel_of(cls, gemm_mode):
    if cls in GEMM_CLASSES:
        return "gemm_split_kernel" if gemm_mode == "split_f16" else "gemm_nt_kernel"
    return cls


def gemm_class_bytes(name, B, J, C, depth=3, T=16, F=2048, streaming=False, gemm_mode="split_f16"):
    """ALGORITHMIC HBM bytes of a GEMM class: every operand and result once (fp32: A, W, C, + residual where there is one)."""
    GH = 1024
    M = B * T * J
    if name == "gemm_lifter":
        per_block = (M * C + 3 * M * C + 3 * C * C) + (M * C + 2 * M * C + C * C) + (M * C + 2 * M * C + 2 * C * C) + (2 * M * C + 2 * M * C + 2 * C * C)
        if C == 256 and gemm_mode == "split_f16":   # proj and fc2 also write their consumer's pre-split LayerNorm output
            per_block += 2 * M * C                  # (pmce_gemm_nt_split_f16_ln: the ln_chain launches' result; the fp32 mode keeps those launches)
        if streaming:
            return 4.0 * (2 * depth - 1) * per_block
        return 4.0 * (B * T * F + B * T * C + F * C + 2 * depth * per_block)
    if name == "gemm_gru_in":
        l1 = 17 * B * 2 * GH + 17 * B * 3 * GH + 6 * GH * 2 * GH
        return 4.0 * (l1 if streaming else 16 * B * F + 16 * B * 6 * GH + 6 * GH * F + l1)
    if name == "gemm_ada":
        return 4.0 * (B * 2048 + B * 3072 + 3072 * 2048)
    if name == "gemm_final":
        return 4.0 * (B * 3360 + B * 20670 + 20670 * 3360)
    return None


def class_work(name, B, J, C, streaming=False):
    """ALGORITHMIC work one forward asks of a kernel class (2*M*N*K of the products actually launched; the pruned GRU
    layer-1 steps and the dead joint stream are NOT counted as work): (amount, unit-kind).  streaming: one WINDOW batch of the
    stride-1 mode (the per-frame products ran once per frame in pmce_stream_precompute and are not in the window pass)."""
    GH, F = 1024, 2048
    if name == "gemm_lifter":
        return gemm_lifter_flops(B, J, C, streaming=streaming), "flop"
    if name == "gemm_gru_in":          # layer 0: 16 steps x 2 directions; layer 1: 9 fwd + 8 bwd steps
        return (0.0 if streaming else 2.0 * 16 * B * 6 * GH * F) + 2.0 * 17 * B * 3 * GH * 2 * GH, "flop"
    if name == "gru_step":             # 16 + 16 layer-0 steps minus the two h0 = 0 steps, 9 + 8 layer-1 steps minus two
        return (30 + 15) * 2.0 * B * 3 * GH * GH, "flop"
    if name == "gemm_final":
        return 2.0 * B * 20670 * 3360, "flop"
    if name == "gemm_ada":
        return 24 * 2 * 2.0 * B * 2048 * 64, "flop"
    if name == "adaln_mlp":
        return 6 * 2 * 2.0 * B * 431 * 64 * 256, "flop"
    if name == "vertex_sa":
        return 3 * (2 * 2.0 * B * 431 * 431 * 64 + 2.0 * B * 431 * 64 * 64), "flop"
    if name == "adaln_qkv":
        return 3 * 2.0 * B * 431 * 64 * 192, "flop"
    if name == "vertex_ca":            # 229,376 B per clip * direction * block
        return 3 * 229376.0 * B, "byte"
    return None, None


def library_build_id():
    """pmce_build_id() of the loaded library (sha over its sources): the id the profiler summaries under profiles/ record."""
    try:
        from pmce_amd import _lib
        return _lib.build_id()
    except Exception:  # noqa: BLE001
        return None


def _pmc_file(fn):
    """(data, meta) of a committed profiler summary; meta carries `stale`: the file does not say it was measured on the build that is
    loaded now (no recorded build id, or another one) - numbers derived from it are flagged, never silently mixed with this run's."""
    path = os.path.join(REPO, "profiles", fn)
    if not os.path.exists(path):
        return None, None
    try:
        data = json.load(open(path))
    except (OSError, ValueError):
        return None, None
    rec_id = (data.get("_meta") or {}).get("build_id")
    lib_id = library_build_id()
    return data, {"file": f"profiles/{fn}", "build_id": rec_id, "library_build_id": lib_id, "stale": not (rec_id and lib_id and rec_id == lib_id)}


def pmc_traffic_per_launch(kernel, C):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC pass of this same command
    (profiles/pmc_hbm_traffic_per_launch_C<C>.json, made by scripts/gpu_session.sh pmc: separate FETCH_SIZE / WRITE_SIZE passes, KiB
    units, FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 correction for wide coalesced reads).  bench.py cannot run
    the profiler on itself; without the file the field is null.  Returns (bytes, source text, meta with the stale flag)."""
    for fn in (f"pmc_hbm_traffic_per_launch_C{C}.json",) + (("pmc_hbm_traffic_per_launch.json",) if C == 256 else ()):
        data, meta = _pmc_file(fn)
        if not data:
            continue
        tot_b, tot_n = 0.0, 0
        for name, v in data.items():
            if name != "_meta" and kernel in name and v.get("launches", 0) > 0:
                tot_b += v["launches"] * (2.0 * v["fetch_kib_raw"] + v["write_kib"]) * 1024.0
                tot_n += v["launches"]
        if tot_n:
            return round(tot_b / tot_n), f"profiles/{fn} (2*FETCH_SIZE + WRITE_SIZE, KiB, launch-weighted)", meta
    return None, None, None


def north_star_record(kernel_ms, launches, B, J, f16_ffn=False, C=512):
    """The CoEvoDecoder vertex<-joint cross-attention (north_star's kernel) against its two floors.  After round 2 it is
    fused with its FFN (`vertex_ca_mlp`); both floors are printed: HBM = SURVEY §8a(a8)'s 229,376 B per clip*direction*block
    at the 8 TB/s peak, MFMA = the matrix instructions the fused kernel must issue (per 32-vertex wave tile: 64 score +
    16*ceil(J/8) output + 512 FFN v_mfma_f32_32x32x2_f32 of 64 cycles each) spread perfectly over the 1024 SIMDs."""
    name = "vertex_ca_mlp" if "vertex_ca_mlp" in kernel_ms else ("vertex_ca" if "vertex_ca" in kernel_ms else None)
    if name is None:
        return None
    n = launches[name]
    ms = kernel_ms[name] / n
    byt = 229376.0 * B
    tiles = B * 14
    # matrix-pipe cycles per 32-vertex wave tile.  fp32 pipe: 64 score + 16 per key group of 8 output + 512 FFN instructions of 64 cycles.
    # split-f16 mode (round 5: the attention's two contractions in the three-product form too): 24 score + 12 per 16 keys output + 192 FFN
    # instructions of 32 cycles.
    if f16_ffn and name == "vertex_ca_mlp":
        cyc_per_tile = (24 + 12 * ((J + 15) // 16) + 192) * 32
    else:
        cyc_per_tile = (64 + 16 * ((J + 7) // 8)) * 64 + (512 * 64 if name == "vertex_ca_mlp" else 0)
    t_hbm = byt / (PEAK_HBM_GBS * 1e9) * 1e3
    t_mfma = tiles * cyc_per_tile / N_SIMD / (CLOCK_GHZ * 1e9) * 1e3
    # vector-pipe floor: the kernel's measured vector instruction count (SQ_INSTS_VALU of the committed PMC pass, per clip), 4
    # cycles each, spread perfectly over the SIMDs - the AdaLN, softmax, exact-erf GELU and operand splitting the kernel cannot
    # avoid in its present form.  A wave issues vector and matrix instructions one after the other, and at 1.75 waves per SIMD
    # little overlaps them: `serial_floor_ms` = matrix + vector time is the realistic floor, max(...) the optimistic one.
    t_valu = None
    pmc, pmc_meta = _pmc_counters(C)
    want = "vertex_ca_mlp_kernel<true" if f16_ffn else "vertex_ca_mlp_kernel<false"      # (template arguments after the first vary with the build)
    key = next((k_ for k_ in pmc if want in k_), None) if pmc else None
    if name == "vertex_ca_mlp" and key and pmc[key].get("SQ_INSTS_VALU") and pmc[key].get("SQ_WAVES"):
        per_clip = pmc[key]["SQ_INSTS_VALU"] / (pmc[key]["SQ_WAVES"] / 7.0)      # 7 waves per clip (14 wave tiles, 2 per wave)
        t_valu = per_clip * B * 4 / N_SIMD / (CLOCK_GHZ * 1e9) * 1e3
    floor = max(t_hbm, t_mfma, t_valu or 0.0)
    ach = byt / (ms * 1e-3) / 1e9
    rec = {"kernel": name, "bound": "hbm" if t_hbm >= max(t_mfma, t_valu or 0.0) else ("mfma" if t_mfma >= (t_valu or 0.0) else "valu"),
           "achieved": round(ach, 1), "peak": PEAK_HBM_GBS,
           "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 4), "bytes_per_clip_dir_block": 229376,
           "avg_launch_ms": round(ms, 5), "hbm_floor_ms": round(t_hbm, 5), "mfma_floor_ms": round(t_mfma, 5),
           "frac_of_floor": round(floor / ms, 4),
           "arithmetic": "attention + FFN: 3 x f16 MFMA per fp32 product" if (f16_ffn and name == "vertex_ca_mlp") else "fp32 MFMA"}
    if t_valu is not None:
        rec.update({"valu_floor_ms": round(t_valu, 5), "serial_floor_ms": round(t_mfma + t_valu, 5),
                    "frac_of_serial_floor": round((t_mfma + t_valu) / ms, 4),
                    # the nominal 4 cycles per wave instruction; plain fp32 FMAs measure 2.4 (profiles/r05_a_f16_mfma_vs_valu_overlap.txt)
                    "valu_floor_at_measured_issue_rate_ms": round(t_valu * 2.4 / 4.0, 5),
                    "note": "f16 matrix instructions and fp32 vector FMAs do not overlap on gfx950 (same file): the serial floor is the realistic one",
                    "valu_source": f"{pmc_meta['file']} (SQ_INSTS_VALU per launch at B = 256, scaled per clip)",
                    "valu_counters_build_id": pmc_meta["build_id"], "library_build_id": pmc_meta["library_build_id"],
                    "stale": pmc_meta["stale"],
                    "mfma_busy_fraction_pmc": pmc[key].get("mfma_busy_fraction"),
                    "wave_cycles_waiting_fraction_pmc": (round(pmc[key]["SQ_WAIT_INST_ANY"] / pmc[key]["SQ_WAVE_CYCLES"], 4)
                                                         if pmc[key].get("SQ_WAVE_CYCLES") and pmc[key].get("SQ_WAIT_INST_ANY") else None)})
    return rec


def _pmc_counters(C=512):
    """SQ counters per kernel (scripts/pmc_kernels.sh) of width C: (data, meta) - meta["stale"] as in _pmc_file."""
    for fn in (f"pmc_counters_per_kernel_C{C}.json",) + (("pmc_counters_per_kernel.json",) if C == 512 else ()):
        data, meta = _pmc_file(fn)
        if data:
            return data, meta
    return {}, None


def mfma_busy_per_variant(kernel, C):
    """Matrix-pipe busy fraction of every instantiation of `kernel` in the committed counter pass:
    (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs)."""
    data, meta = _pmc_counters(C)
    out = {n: v["mfma_busy_fraction"] for n, v in data.items() if n != "_meta" and kernel in n and isinstance(v, dict) and "mfma_busy_fraction" in v}
    return (out, meta) if out else (None, meta)


def attention_record(kernel_ms, launches, B, J, C, split):
    """The lifter's attention - the HBM-bound kernel of the path - against the HBM roofline: it reads q, k, v (3C floats per
    token) and writes C floats per token, once each; `traffic` is what the PMC pass counted per launch."""
    if "seq_attention" not in kernel_ms or not launches.get("seq_attention"):
        return None
    ms = kernel_ms["seq_attention"] / launches["seq_attention"]
    byt = B * 16 * J * 4 * C * 4.0
    ach = byt / (ms * 1e-3) / 1e9
    kern = "seq_attention_mfma_kernel" if split else ("seq_attention_pair_kernel" if C == 512 else "seq_attention_kernel")
    tr, src, tmeta = pmc_traffic_per_launch(kern, C)
    return {"kernel": kern, "bound": "hbm", "traffic_stale": (tmeta or {}).get("stale"), "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": round(ach / PEAK_HBM_GBS, 4), "frac_of_achievable_6300": round(ach / 6300.0, 4), "avg_launch_ms": round(ms, 5),
            "launches_per_step": launches["seq_attention"], "algorithmic_bytes_per_launch": int(byt), "traffic": tr,
            "traffic_source": src}


def dominant_kernel_roofline(kernel_ms, launches, B, J, C, gemm_mode, clk_ghz=None, streaming=False):
    """`roofline` of a profiled configuration: the kernel (function, not timing class) with the largest HIP-event time, its
    algorithmic work per launch against the peak that bounds it.  kernel_ms / launches: per timing class of model.cpp, per step."""
    by_kernel = {}
    for k_ in kernel_ms:
        by_kernel.setdefault(kernel_of(k_, gemm_mode), []).append(k_)
    # the kernel with the largest time among those whose algorithmic work is modelled (class_work): at tiny batches a latency-bound kernel
    # without a work model (joint_stream, ca_fold) can lead - it is named in `larger_unmodelled_kernels`, the roofline is the next one's
    ranked = sorted(by_kernel, key=lambda kn: -sum(kernel_ms[c] for c in by_kernel[kn]))
    modelled = [kn for kn in ranked if all(class_work(c, B, J, C, streaming)[0] is not None for c in by_kernel[kn])]
    if not modelled:
        return None
    dominant = modelled[0]
    unmodelled_ahead = ranked[:ranked.index(dominant)]
    dom_classes = by_kernel[dominant]
    dom_ms = sum(kernel_ms[c] for c in dom_classes)
    dom_launches = sum(launches[c] for c in dom_classes)
    works = [class_work(c, B, J, C, streaming) for c in dom_classes]
    roofline = None
    if all(w[0] is not None for w in works):
        kind = works[0][1]
        work = sum(w[0] for w in works)
        secs = dom_ms * 1e-3
        traffic, traffic_src, tmeta = pmc_traffic_per_launch(dominant, C)
        busy, bmeta = mfma_busy_per_variant(dominant, C)
        common = {"kernel": dominant, "classes": dom_classes, "launches_per_step": dom_launches,
                  "avg_launch_ms": round(dom_ms / dom_launches, 5), "traffic": traffic, "traffic_source": traffic_src,
                  "traffic_stale": (tmeta or {}).get("stale"), "library_build_id": library_build_id(),
                  "mfma_busy": busy, "mfma_busy_source": (bmeta or {}).get("file"), "mfma_busy_stale": (bmeta or {}).get("stale"),
                  "algorithmic_per_launch": work / dom_launches}
        if unmodelled_ahead:
            common["larger_unmodelled_kernels"] = unmodelled_ahead
        if kind == "flop" and dominant == "gemm_split_kernel":
            # the kernel issues THREE f16 matrix products per algorithmic fp32 product: achieved = issued f16 FLOP/s against the
            # dense f16 peak; the fp32-equivalent rate and the same launches against the HBM roofline (every operand and
            # result once) are printed beside it - at these shapes the two floors are within 15 % of each other
            # the kernel issues THREE f16 matrix products per algorithmic fp32 product.  `achieved` / `frac` are the ALGORITHMIC rate
            # (2MNK of the fp32 products the reference asks for) against the dense f16 peak of the pipe it runs on; the
            # rate of the f16 FLOPs it issues (3 x 2MNK) is carried beside it as `achieved_issued` / `frac_issued`
            ach = work / secs / 1e12
            iss = 3.0 * ach
            byt = sum(gemm_class_bytes(c, B, J, C, streaming=streaming, gemm_mode=gemm_mode) for c in dom_classes)
            roofline = {"bound": "mfma", "achieved": round(ach, 1), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(ach / PEAK_F16_TFLOPS, 4), **common,
                        "arithmetic": "3 x v_mfma_f32_32x32x16_f16 per fp32 product (hi*hi + hi*lo + lo*hi), fp32 accumulate",
                        "achieved_issued": round(iss, 1), "frac_issued": round(iss / PEAK_F16_TFLOPS, 4),
                        "measured_pipe_ceiling_tflops": MEASURED_F16_CEILING_TFLOPS,
                        "frac_issued_of_measured_pipe_ceiling": round(iss / MEASURED_F16_CEILING_TFLOPS, 4),
                        "note": "achieved / frac = algorithmic 2MNK of the fp32 products per second against the dense f16 peak; "
                                "*_issued counts the three f16 products the kernel issues per fp32 product",
                        # MI355X runs this kernel power-limited: the shader clock its workgroups measured (s_memtime vs the 100 MHz
                        # wall counter, in these very launches), and the issued rate against the matrix peak AT that clock
                        "sustained_clock_ghz": round(clk_ghz, 3) if clk_ghz else None,
                        "frac_issued_of_peak_at_sustained_clock": round(iss / (PEAK_F16_TFLOPS * clk_ghz / 2.4), 4) if clk_ghz else None,
                        "hbm": {"algorithmic_bytes_per_launch": round(byt / dom_launches), "achieved_gbs": round(byt / secs / 1e9, 1),
                                "peak_gbs": PEAK_HBM_GBS, "frac": round(byt / secs / 1e9 / PEAK_HBM_GBS, 4)}}
        elif kind == "flop":
            ach = work / secs / 1e12
            roofline = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(ach / PEAK_F32_TFLOPS, 4), **common}
        else:
            ach = work / secs / 1e9
            roofline = {"bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": round(ach / PEAK_HBM_GBS, 4), **common}
    return roofline


# ------------------------------------------------------------------------------------------------------------------
def measure_config(args, dev, rank, world, J, C, B, steps, warmup, windows, full=True):
    """Throughput + per-kernel-class timing of one configuration.  Returns (record, model, pipe, inputs)."""
    import torch
    from pmce_amd import _lib, assets, models, sharding, synth
    from pmce_amd.workload import flops_per_clip

    assets.allow_synthetic_base_data()                                  # no SMPL-derived files offline: synthetic template
    t_load0 = time.perf_counter()
    # random-init weights of the named architecture; at N > 1 generated once per node and mapped by the other ranks
    if world > 1:
        nonce = int(sharding.reduce_max(float(os.getpid() if rank == 0 else 0), dev))      # rank 0's pid, known to every rank
        sd, shm_path = shared_state_dict(J, C, int(os.environ.get("LOCAL_RANK", rank)), sharding.barrier, nonce)
    else:
        sd, shm_path = synth.make_state_dict(synth.pmce_spec(J, C, 3), seed=123), None
    model = models.PMCE.get_model(J, C, 3)
    model.load_state_dict(sd)
    model.set_j_regressor(assets.load_j_regressor("h36m"))
    model = model.to(dev)
    torch.cuda.synchronize()
    weights_load_s = time.perf_counter() - t_load0                       # this rank: weights generated / mapped, loaded, on its GPU
    if world > 1:
        sharding.barrier()                                               # every rank has its copy on its GPU
        if shm_path and int(os.environ.get("LOCAL_RANK", rank)) == 0:
            try:
                os.remove(shm_path)
            except OSError:
                pass
    model.set_gemm_mode(args.gemm_mode)
    gemm_mode = model.gemm_mode()
    # throughput loops launch ahead of the GPU: the module's default policy ("rerun": forward() waits for its result and polls the overflow
    # word) would serialise host and device.  "report" keeps calls asynchronous; the outputs are checked finite after the timed region.
    model.set_overflow_policy("report")

    # synthetic clips resident in HBM; NB distinct batches are rotated so that the timed loop's inputs (NB x 33.6 MB at
    # B = 256) do not sit in the 256 MB Infinity Cache from one step to the next
    NB = max(1, min(8, (288 << 20) // max(1, B * (16 * 2048 + 16 * J * 2) * 4) + 1))
    gen = torch.Generator(device=dev)
    gen.manual_seed(1000 + rank)
    inputs = []
    for _ in range(NB):
        p = torch.rand(B, 16, J, 2, device=dev, generator=gen) * 2 - 1                 # normalised screen coordinates
        f = torch.relu(torch.randn(B, 16, 2048, device=dev, generator=gen))            # non-negative pooled CNN features
        inputs.append((p, f))

    depth = 1 if args.single_stream else max(1, args.pipeline_depth)
    pipe = model.pipeline(depth, stagger=(True if args.stagger else False if args.no_stagger else None)).prepare(B) if depth > 1 else None
    if args.single_stream:
        model.set_concurrency(False)
    k = [0]

    def step(direct=False):
        p, f = inputs[k[0] % NB]
        k[0] += 1
        if pipe is None or direct:
            return model.forward_with_joints(p, f)
        return pipe.submit(p, f)     # returns once the batch is enqueued on its lane; complete before the closing synchronize

    out = None
    t_first0 = time.perf_counter()
    for i in range(max(warmup, 1)):  # packing / workspace allocation must not be inside the timed region
        out = step()
        if i == 0:
            torch.cuda.synchronize()
            first_step_s = time.perf_counter() - t_first0                # this rank's first step: weight packing, workspace, kernel load
    win_ms, own_dt = [], []
    for _ in range(windows):
        torch.cuda.synchronize()
        sharding.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step()
        torch.cuda.synchronize()
        own_dt.append(time.perf_counter() - t0)      # this rank's own work (before it waits for the others)
        sharding.barrier()
        torch.cuda.synchronize()
        dt = sharding.reduce_max(time.perf_counter() - t0, dev)
        win_ms.append(dt / steps * 1e3)
    ms = statistics.median(win_ms)
    clips_per_s = B * world / (ms * 1e-3)
    # every rank's OWN rate over the last window (its own clock, before the MAX): a straggler is visible here
    per_rank = sharding.gather_rows(torch.tensor([[B / (own_dt[-1] / steps)]], dtype=torch.float64, device=dev)).flatten().tolist()
    startup = sharding.gather_rows(torch.tensor([[weights_load_s, first_step_s]], dtype=torch.float64, device=dev)).tolist()

    # final metric reduction — the only collective of the path (RCCL over xGMI): per-rank partial sums
    if pipe is not None:
        out = out.result()
    mesh, pose, pose3d, pred = out
    partial = torch.stack([pred.abs().sum().double(), mesh.abs().sum().double(),
                           torch.tensor(float(B), device=dev, dtype=torch.float64)])
    total = sharding.reduce_metric_sums(partial)
    finite = bool(torch.isfinite(total).all().item()) and int(total[2].item()) == B * world

    fpc = flops_per_clip(J, C)["total"]
    rec = {"value": round(clips_per_s, 1), "unit": "clips/s", "ms_per_step": round(ms, 4),
           "windows": {"n": windows, "steps_each": steps, "ms_per_step": [round(x, 4) for x in win_ms],
                       "value_min": round(B * world / (max(win_ms) * 1e-3), 1),
                       "value_max": round(B * world / (min(win_ms) * 1e-3), 1)},
           "config": {"workload": f"full two-stream PMCE forward (temporal pose encoder + CoEvoDecoder + 6890-vertex upsample + "
                                  f"J_regressor), batch={B}/GPU, T=16, J={J}, C={C}, random-init weights",
                      "global_batch": B * world, "seq_len": 16, "joints": J, "embed_dim": C,
                      "parallelism": f"clip-sharded dp{world}, weights replicated",
                      "gemm_mode": gemm_mode, "overflow_policy": "report (asynchronous calls; outputs_finite is checked after the timed region)",
                      "streams": 1 if (args.single_stream or (gemm_mode == "split_f16" and not _lib.split_overlap())) else 2 * depth,
                      "batches_enqueued_ahead": depth, "lanes": ("staggered" if pipe.staggers(B) else "free-running") if pipe is not None else "one batch at a time",
                      "distinct_input_batches": NB},
           "ref_equiv_tflops": round(fpc * clips_per_s / 1e12, 2), "outputs_finite": finite,
           "per_rank_clips_s": [round(x, 1) for x in per_rank],
           "per_rank_ms_per_step": [round(B / x * 1e3, 4) for x in per_rank],
           # start-up skew of the ranks (not in any timed window): weights generated or mapped + loaded + on the GPU; the first step
           "per_rank_weights_load_s": [round(r[0], 3) for r in startup], "per_rank_first_step_s": [round(r[1], 3) for r in startup],
           "metric_reduction": {"collective": "all_reduce(SUM) of 3 fp64 partials per rank", "clips_counted": int(total[2].item()),
                                "backend": (torch.distributed.get_backend() if world > 1 else None)}}
    if args.sustained_seconds > 0 and not args.single_stream:
        rec["sustained"] = sustained_record(model, pipe, step, dev, B, world, args.sustained_seconds, clips_per_s, gemm_mode)
    if not full:
        return rec, model, pipe, inputs, sd

    # ---- per-kernel-class timing with HIP events on the launch stream (a few extra, untimed-for-value steps, one batch
    # at a time on one stream so that every launch is priced alone) ----
    run1 = lambda: model.forward_with_joints(*inputs[0])
    clk = torch.zeros(2, dtype=torch.int64, device=dev)
    lib = _lib.load()
    if gemm_mode == "split_f16":       # the split kernel's workgroups report their residence in shader clocks and 100 MHz ticks
        lib.pmce_model_set_clock_probe(model._ensure_packed().handle, ctypes.c_void_p(clk.data_ptr()))
    model.profile(True)
    nprof = 3
    for _ in range(nprof):
        run1()
    torch.cuda.synchronize()
    lib.pmce_model_set_clock_probe(model._ensure_packed().handle, None)
    prof = model.profile_read()
    model.profile(False)
    clk_ghz = float(clk[0].item()) / float(clk[1].item()) * 0.1 if int(clk[1].item()) > 0 else None
    if args.single_stream:
        model.set_concurrency(False)
    kernel_ms = {k_: round(v[0] / nprof, 4) for k_, v in prof.items() if v[1] > 0}
    launches = {k_: int(v[1] // nprof) for k_, v in prof.items() if v[1] > 0}
    roofline = dominant_kernel_roofline(kernel_ms, launches, B, J, C, gemm_mode, clk_ghz)
    # every rank's own sustained shader clock under the dominant kernel (power-limited: 8 GPUs in one chassis need not hold the same)
    rec["per_rank_sustained_clock_ghz"] = [round(x, 3) for x in
                                           sharding.gather_rows(torch.tensor([[clk_ghz or 0.0]], dtype=torch.float64, device=dev)).flatten().tolist()]
    rec.update({"roofline": roofline, "roofline_cross_attention": north_star_record(kernel_ms, launches, B, J, f16_ffn=(gemm_mode == "split_f16"), C=C),
                "roofline_attention": attention_record(kernel_ms, launches, B, J, C, gemm_mode == "split_f16" and J in (17, 19)),
                "kernel_ms_per_step": kernel_ms, "launches_per_step": launches,
                "kernel_ms_total_single_stream": round(sum(kernel_ms.values()), 4), "gemm_mode": gemm_mode})
    return rec, model, pipe, inputs, sd


def sustained_record(model, pipe, step, dev, B, world, seconds, burst_value, gemm_mode):
    """BASELINE config 5 asks for SUSTAINED clips/s and every large kernel of the path is power-limited: `value` is the median of a
    few windows of well under a second each.  This is ONE window of at least `seconds` of back-to-back steps on the same workload (same
    batches in flight, same rotating inputs), with the shader clock the split GEMM's workgroups measure (s_memtime against the 100 MHz
    counter) sampled at the start, in the middle and at the end - three single forwards on the model's own handle with the clock probe
    on, counted as steps.  Reported next to `value`, never as it."""
    import torch
    from pmce_amd import _lib, sharding
    lib = _lib.load()
    handle = model._ensure_packed().handle
    clk = torch.zeros(2, dtype=torch.int64, device=dev)

    def probe():
        if gemm_mode != "split_f16":
            return None
        if pipe is not None:
            pipe.synchronize()
        clk.zero_()
        torch.cuda.synchronize()
        lib.pmce_model_set_clock_probe(handle, ctypes.c_void_p(clk.data_ptr()))
        step(direct=True)
        torch.cuda.synchronize()
        lib.pmce_model_set_clock_probe(handle, None)
        c = clk.tolist()
        return round(c[0] / c[1] * 0.1, 3) if c[1] > 0 else None

    def smi():
        try:
            raw = float(torch.cuda.power_draw())      # documented as mW; torch 2.10 on ROCm returns W (1,100 - 1,400 here)
            return {"power_w": raw if raw < 5000 else raw / 1000.0, "temp_c": torch.cuda.temperature()}
        except Exception:  # noqa: BLE001
            return None

    torch.cuda.synchronize()
    sharding.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    clocks, smis, marks = [], [], []
    for phase_end in (0.0, seconds / 2, seconds):
        while time.perf_counter() - t0 < phase_end:
            for _ in range(8):
                step()
            n += 8
            if n % 64 == 0:
                torch.cuda.synchronize()       # bound the launch queue (a few hundred ms of work at most)
        marks.append(round(time.perf_counter() - t0, 2))
        smis.append(smi())
        clocks.append(probe())
        n += 1 if clocks[-1] is not None else 0
    torch.cuda.synchronize()
    if pipe is not None:
        pipe.synchronize()
    own = time.perf_counter() - t0
    sharding.barrier()
    dt = sharding.reduce_max(time.perf_counter() - t0, dev)
    val = B * world * n / dt
    return {"value": round(val, 1), "unit": "clips/s", "seconds": round(dt, 2), "steps": n, "ms_per_step": round(dt / n * 1e3, 4),
            "this_rank_seconds": round(own, 2),
            "clock_ghz_start": clocks[0], "clock_ghz_middle": clocks[1], "clock_ghz_end": clocks[2], "clock_sample_at_s": marks,
            "device_power_temp": smis if any(smis) else None,
            "ratio_to_value": round(val / burst_value, 4) if burst_value else None,
            "what": "one window of back-to-back steps (same batches in flight and rotating inputs as `value`); clock = shader clock measured by "
                    "the split GEMM's workgroups during one forward at the start / middle / end"}


def host_fed_record(model, pipe, dev, B, J, nfed):
    """The same clips start in pageable host memory every step and travel through pmce_amd.staging.PinnedFeeder (memcpy to
    a pinned ring, async H2D on a copy stream under the previous step's kernels).  Reported next to `value`, never as it."""
    import torch
    from pmce_amd import staging, synth
    pose2d_np, feat_np = synth.make_inputs(B, J, seed=1000)
    feeder = staging.PinnedFeeder(dev, {"pose2d": ((B, 16, J, 2), torch.float32), "img_feat": ((B, 16, 2048), torch.float32)}, slots=4)
    it = feeder.run([{"pose2d": pose2d_np, "img_feat": feat_np}] * (nfed + 2))

    def consume(d):
        if pipe is None:
            model.forward_with_joints(d["pose2d"], d["img_feat"])
        else:                                    # two batches in flight here too: the slot is free when ITS lane is done
            d.release(pipe.submit(d["pose2d"], d["img_feat"]).done)

    for _ in range(2):                           # warm the pinned ring
        consume(next(it))
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for d in it:
        consume(d)
    torch.cuda.synchronize()
    t_fed = time.perf_counter() - t1
    return {"value": round(B * nfed / t_fed, 1), "unit": "clips/s", "steps": nfed,
            "h2d_bytes_per_step": int(pose2d_np.nbytes + feat_np.nbytes),
            "path": "pageable host -> pinned ring (4 slots) -> async H2D on a copy stream -> forward (same batches in flight as `value`)"}


def latency_record(model, dev, J, calls=200):
    """Small-batch latency (the reference's demo runs the path at batch 1, main/run_demo.py:332,145): host-observed time
    of one forward_with_joints + synchronize, p50/p99 over `calls` calls, eager launches and hipGraph replay."""
    import torch
    out = {}
    for B in (1, 8):
        p = torch.rand(B, 16, J, 2, device=dev) * 2 - 1
        f = torch.relu(torch.randn(B, 16, 2048, device=dev))
        rec = {}
        for mode in ("eager", "graph"):
            try:
                if mode == "eager":
                    run = lambda: model.forward_with_joints(p, f)
                else:
                    gf = model.graphed(B)
                    run = lambda: gf(p, f)
                for _ in range(10):
                    run()
                torch.cuda.synchronize()
                ts = []
                for _ in range(calls):
                    t0 = time.perf_counter()
                    run()
                    torch.cuda.synchronize()
                    ts.append((time.perf_counter() - t0) * 1e3)
                ts.sort()
                rec[mode] = {"p50_ms": round(ts[len(ts) // 2], 4), "p99_ms": round(ts[min(len(ts) - 1, int(len(ts) * 0.99))], 4),
                             "min_ms": round(ts[0], 4)}
            except Exception as e:  # noqa: BLE001
                rec[mode] = {"error": str(e)[:200]}
        # where a single small forward spends its time: HIP events around every launch, one stream (3 forwards)
        model.profile(True)
        for _ in range(3):
            model.forward_with_joints(p, f)
        torch.cuda.synchronize()
        prof = model.profile_read()
        model.profile(False)
        rec["kernel_ms_single_stream"] = {k: round(v[0] / 3, 4) for k, v in prof.items() if v[1] > 0 and v[0] / 3 >= 0.01}
        rec["kernel_ms_total_single_stream"] = round(sum(v[0] for v in prof.values()) / 3, 4)
        out[f"B{B}"] = rec
    out["calls"] = calls
    out["what"] = "forward_with_joints + device synchronize, host clock"
    return out


def cpu_baseline_record(sd, vj_relation, J, C, seconds):
    """The oracle (a port of the reference forward: kind "port") on this box's host cores — a bounded sample."""
    import torch
    from oracle import pmce_oracle as O
    from pmce_amd import synth
    ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)   # (a rank's own CPU slice at N > 1)
    # torch's intra-op pool degrades badly when oversubscribed on many-core hosts: calibrate the thread count on a small
    # batch (a few seconds), then time a bounded sample of batch-64 forwards with the best one.
    p_cpu, f_cpu = (torch.from_numpy(a) for a in synth.make_inputs(64, J, seed=1))
    best_t, best_rate = 1, 0.0
    cal = {}
    with torch.no_grad():
        torch.set_num_threads(min(16, ncores))
        t1 = time.perf_counter()
        O.pmce_forward(sd, p_cpu[:4], f_cpu[:4], vj_relation)
        cb = 64 if 4 / (time.perf_counter() - t1) > 4 else 16      # (a very slow host times batch-16 forwards instead)
        for nt in [t for t in (8, 16, 32, 64, 128, 256) if t <= ncores] or [ncores]:   # calibrated on the batch that is timed
            torch.set_num_threads(nt)
            O.pmce_forward(sd, p_cpu[:2], f_cpu[:2], vj_relation)  # warm the pool
            t1 = time.perf_counter()
            O.pmce_forward(sd, p_cpu[:cb], f_cpu[:cb], vj_relation)
            rate = cb / (time.perf_counter() - t1)
            cal[nt] = round(rate, 1)
            if rate > best_rate:
                best_t, best_rate = nt, rate
            elif rate < 0.6 * best_rate:     # past the knee: more threads only oversubscribe (256 threads: 0.3 clips/s, minutes per forward)
                break
        torch.set_num_threads(best_t)
        n, t_cpu = 0, 0.0
        while t_cpu < seconds and n < 50:
            t1 = time.perf_counter()
            O.pmce_forward(sd, p_cpu[:cb], f_cpu[:cb], vj_relation)
            t_cpu += time.perf_counter() - t1
            n += 1
        lat = []
        for _ in range(5):                       # the demo's batch: one clip
            t1 = time.perf_counter()
            O.pmce_forward(sd, p_cpu[:1], f_cpu[:1], vj_relation)
            lat.append((time.perf_counter() - t1) * 1e3)
    return {"value": round(cb * n / t_cpu, 2), "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
            "host_logical_cpus": ncores, "ms_per_clip": round(t_cpu / (cb * n) * 1e3, 3),
            "batch1_latency_ms": round(statistics.median(lat), 2),
            "threads_calibration_clips_s": cal,
            "sample": f"{n} x batch-{cb} full forwards of oracle/pmce_oracle.py (torch CPU fp32, J={J}, C={C}), thread count "
                      f"calibrated over 8, 16, 32, ... on batch-{cb} forwards (the batch that is timed) until the rate falls below 0.6 of the best; batch1_latency_ms = median of 5 "
                      f"single-clip forwards"}


# ------------------------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--windows", type=int, default=5, help="timed windows of --steps steps each; `value` is the median window")
    ap.add_argument("--batch", type=int, default=256, help="clips per GPU per step")
    ap.add_argument("--joints", type=int, default=17)
    ap.add_argument("--embed-dim", type=int, default=512,
                    help="pose-encoder width of the headline record: 512 = BASELINE.json north_star; 256 = the reference's shipped width")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--single-stream", action="store_true",
                    help="keep every launch on one stream (profiling: rocprofv3 then prices each kernel alone)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--pipeline-depth", type=int, default=2,
                    help="batches in flight (models.PMCE.Pipeline): a step's decoder overlaps the next step's pose lifter; "
                         "1 = strictly one batch at a time")
    ap.add_argument("--stagger", action="store_true", help="always start a batch's lifter when the previous batch's has finished (default: the "
                                                           "pipeline decides from the batch size - staggered from 192 clips on)")
    ap.add_argument("--no-stagger", action="store_true", help="always let the pipeline lanes run free")
    ap.add_argument("--sustained-seconds", type=float, default=20.0,
                    help="length of the one long window behind the `sustained` record (0 = skip); the C = 256 record runs half of it")
    ap.add_argument("--no-host-fed", action="store_true", help="skip the extra (untimed-for-value) host-fed measurement")
    ap.add_argument("--no-latency", action="store_true", help="skip the B=1 / B=8 latency record")
    ap.add_argument("--no-variant", action="store_true", help="skip the second complete record (the other pose-encoder width)")
    ap.add_argument("--gemm-mode", choices=("split_f16", "f32"), default="split_f16",
                    help="the large products as three f16 matrix products each (fp32 accumulate, fp32 accuracy) or on the fp32 matrix "
                         "pipe; both with two streams inside a forward and two batches in flight")
    ap.add_argument("--detail-file", default=os.path.join(REPO, "bench_detail.json"),
                    help="where rank 0 writes the FULL record (variants, other BASELINE configs, latency, per-kernel tables); stdout carries "
                         "only the compact record, as its one and last line")
    ap.add_argument("--dist-check", action="store_true",
                    help="rendezvous + the path's collectives only (no GPU work): what the non-GPU test of the N>1 entry point runs")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    # stdout carries the record and nothing else: libraries write there too (gloo prints "[Gloo] Rank 0 is connected ..." on stdout, RCCL its
    # NCCL_DEBUG lines).  File descriptor 1 is pointed at stderr for the rest of the process; the record goes to a private copy of the real one.
    global _RECORD_OUT
    sys.stdout.flush()
    _RECORD_OUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    # The second complete record (the other pose-encoder width) is measured by a CHILD process running this same script,
    # BEFORE this process touches the GPU: a second model in one process shares HIP's few hardware queues with the first
    # one's (pooled, never destroyed) streams and loses the two-batches overlap (29.2 k instead of 32.1 k clips/s at
    # C = 256), and a child that runs while its parent holds a HIP context sees ~6 % slower kernels in its profiling pass.
    variant = variant_f32 = None
    C, B, J = args.embed_dim, args.batch, args.joints
    keep = ("value", "unit", "ms_per_step", "windows", "sustained", "config", "roofline", "roofline_cross_attention", "roofline_attention", "cpu_baseline",
            "kernel_ms_per_step", "launches_per_step", "kernel_ms_total_single_stream", "ref_equiv_tflops", "outputs_finite")

    def child_record(extra, cpu_ok, sus=0.0):
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(args.steps), "--warmup", str(args.warmup),
               "--windows", str(args.windows), "--batch", str(B), "--joints", str(J),
               "--pipeline-depth", str(args.pipeline_depth), "--no-variant", "--no-host-fed", "--no-latency",
               "--cpu-seconds", str(min(args.cpu_seconds, 8.0)), "--sustained-seconds", str(sus), *extra]
        cmd += ["--stagger"] if args.stagger else ["--no-stagger"] if args.no_stagger else []
        cmd += ["--no-cpu-baseline"] if (args.no_cpu_baseline or not cpu_ok) else []
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
        import tempfile
        with tempfile.NamedTemporaryFile(suffix=".json") as tf:
            r = subprocess.run([*cmd, "--detail-file", tf.name], env=env, capture_output=True, text=True)
            try:
                d = json.load(open(tf.name)) if r.returncode == 0 else None
            except ValueError:
                d = None
        if d:
            rec = {k: d.get(k) for k in keep}
            rec["measured_by"] = "child process running this same script, before this process touched the GPU"
            return rec
        return {"error": (r.stderr or r.stdout)[-400:]}

    def script_record(script, extra, what):
        """One of BASELINE.json's other configurations, run by its own script in a child process (same reasons as above); the
        script prints one JSON line with the configuration's rate and the roofline of its dominant kernel."""
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
        r = subprocess.run([sys.executable, os.path.join(REPO, "scripts", script), *extra], env=env, capture_output=True, text=True)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode == 0 and lines:
            d = json.loads(lines[-1])
            d["what"] = what
            d["measured_by"] = f"child process: python scripts/{script} {' '.join(extra)}"
            return d
        return {"error": (r.stderr or r.stdout)[-400:], "what": what}

    other_configs = {}
    if (args.gpus == 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_variant and not args.single_stream
            and not args.dist_check and C in (256, 512)):
        # BASELINE.json configs[1], [3], [4] (C = 256, the width every reference config ships): decoder-only at batch 64; the
        # 3DPW-sized clip-sharded evaluation (35,515 clips, COCO input J = 19: forward + on-device metrics + the one reduction)
        # and the stride-1 streaming of one long sequence (frame reuse + acceleration error), both on synthetic stand-ins
        other_configs["config_decoder_b64"] = script_record(
            "decoder_bench.py", ["--batch", "64", "--steps", "50", "--min-seconds", "5"], "BASELINE configs[1]: CoEvoDecoder-only forward, batch 64, 1 GPU")
        other_configs["config_eval_sharded_j19"] = script_record(
            "eval_sharded.py", ["--clips", "35515", "--joints", "19", "--min-seconds", "5"],
            "BASELINE configs[3] stand-in: 35,515 clips (3DPW test-set size), J = 19, forward + on-device MPJPE / PA-MPJPE / MPVPE / accel + "
            "the final reduction; this rank count's shard of the clip range")
        other_configs["config_streaming"] = script_record(
            "stream_bench.py", ["--frames", "16384", "--min-seconds", "5"],
            "BASELINE configs[4] stand-in: one 16,384-frame sequence, stride-1 T = 16 windows with per-frame reuse, acceleration error on the device")
        C2 = 256 if C == 512 else 512
        variant = child_record(["--embed-dim", str(C2), "--gemm-mode", args.gemm_mode], True, sus=args.sustained_seconds / 2)
        variant["why"] = ("the width every reference config ships (lib/core/config.py:59)" if C2 == 256
                          else "BASELINE.json north_star's width")
        if args.gemm_mode == "split_f16":     # the same workload with every product on the fp32 matrix pipe
            variant_f32 = child_record(["--embed-dim", str(C), "--gemm-mode", "f32"], False)
            variant_f32["why"] = ("every product on v_mfma_f32_32x32x2_f32 (157 TFLOP/s peak), two streams and two batches in "
                                  "flight: the round-1 arithmetic, for comparison")

    import torch
    from pmce_amd import sharding

    rank, local, world = sharding.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}")
    cpu_slice = bind_rank_to_cpus(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)))

    if args.dist_check:
        import torch.distributed as dist
        cdev = torch.device("cpu")
        lo, hi = sharding.shard_range(1000, rank, world)
        tot = sharding.reduce_metric_sums(torch.tensor([float(hi - lo), 1.0], dtype=torch.float64))
        rows = sharding.gather_rows(torch.arange(lo, hi, dtype=torch.float32)[:, None])
        tmax = sharding.reduce_max(float(rank), cdev)
        sharding.barrier()
        if rank == 0:
            _print_record(json.dumps({"dist_check": True, "n_gpus": world, "backend": dist.get_backend() if world > 1 else None,
                              "clips": int(tot[0].item()), "ranks": int(tot[1].item()), "gathered": int(rows.shape[0]),
                              "max_rank": int(tmax)}))
        if world > 1:
            dist.destroy_process_group()
        return

    ndev = torch.cuda.device_count()
    if os.environ.get("PMCE_BENCH_SHARE_GPU"):   # plumbing runs of the N>1 path on a box with fewer GPUs (ranks share devices)
        local = local % max(ndev, 1)
    if local >= ndev:
        raise SystemExit(f"bench.py: rank {rank} wants cuda:{local} but only {ndev} GPU(s) are visible "
                         f"(set PMCE_BENCH_SHARE_GPU=1 PMCE_DIST_BACKEND=gloo to let ranks share a device)")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if world > 1 and not os.environ.get("PMCE_BENCH_SHARE_GPU"):
        import torch.distributed as dist
        if dist.get_backend() != "nccl":      # one rank per GPU: the metric reduction must run over RCCL / xGMI, never silently over gloo
            raise SystemExit(f"bench.py: {world} ranks on {ndev} GPU(s) but the process group backend is {dist.get_backend()!r}, not 'nccl' "
                             f"(RCCL); unset PMCE_DIST_BACKEND")

    head, model, pipe, inputs, sd = measure_config(args, dev, rank, world, J, C, B, args.steps, args.warmup, args.windows)

    solo = rank == 0 and world == 1
    host_fed = latency = cpu = None
    if solo and not args.no_host_fed and not args.single_stream:
        host_fed = host_fed_record(model, pipe, dev, B, J, max(5, min(args.steps, 20)))
    if solo and not args.no_latency:
        latency = latency_record(model, dev, J)
    vj = model.vj_relation
    del model, pipe, inputs
    torch.cuda.empty_cache()

    if rank == 0 and not args.no_cpu_baseline:      # host-only work last; at N > 1 on rank 0 while the others wait at the final barrier
        cpu = cpu_baseline_record(sd, vj, J, C, args.cpu_seconds)

    if rank == 0:
        line = {
            "metric": "16-frame clips/s", "value": head["value"], "unit": "clips/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32",
            "dtype_note": ("fp32 operands, results and accumulation; in split_f16 mode each large fp32 product is three f16 MFMA products "
                           "(hi*hi + hi*lo + lo*hi), error vs fp64 at or below the fp32 matrix pipe's"),
            "data": "synthetic",
            "config": head["config"], "roofline": head["roofline"], "roofline_cross_attention": head["roofline_cross_attention"],
            "roofline_attention": head.get("roofline_attention"),
            "cpu_baseline": cpu, "windows": head["windows"], "sustained": head.get("sustained"), "host_fed": host_fed, "latency": latency,
            f"variant_c{256 if C == 512 else 512}": variant, "variant_f32_pipe": variant_f32,
            "kernel_ms_per_step": head["kernel_ms_per_step"], "launches_per_step": head["launches_per_step"],
            "kernel_ms_total_single_stream": head["kernel_ms_total_single_stream"],
            "ref_equiv_tflops": head["ref_equiv_tflops"], "outputs_finite": head["outputs_finite"],
            "per_rank_clips_s": head["per_rank_clips_s"], "per_rank_ms_per_step": head["per_rank_ms_per_step"],
            "per_rank_sustained_clock_ghz": head.get("per_rank_sustained_clock_ghz"),
            "per_rank_weights_load_s": head.get("per_rank_weights_load_s"), "per_rank_first_step_s": head.get("per_rank_first_step_s"),
            "library_build_id": library_build_id(),
            "rank0_cpu_affinity": ({"cpus": len(cpu_slice), "first": cpu_slice[0], "last": cpu_slice[-1]} if cpu_slice else None),
            "metric_reduction": head["metric_reduction"],
            **other_configs,
        }
        emit(line, args.detail_file)
    if world > 1:
        import torch.distributed as dist
        sharding.barrier()        # (rank 0 timed the CPU baseline meanwhile)
        dist.destroy_process_group()


_RECORD_OUT = None       # the process's real stdout (main() points fd 1 at stderr so that library chatter cannot share a stream with the record)


def _print_record(text: str):
    out = _RECORD_OUT or sys.stdout
    out.write(text + "\n")
    out.flush()


COMPACT_LIMIT = 4096     # bytes; the round driver keeps only a few KB of stdout (BENCH_r05.json: a 24.7 KB line came back unparsed)


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d} if isinstance(d, dict) else None


def compact_record(full: dict, detail_file: str | None) -> dict:
    """The ONE line bench.py prints: the contract's fields, `roofline` of the dominant kernel (frac = algorithmic), `cpu_baseline`,
    the sustained window and one headline number per other record.  Everything else lives in the detail file."""
    rf = full.get("roofline") or {}
    src = rf.get("traffic_source")
    roofline = _pick(rf, ("bound", "kernel", "achieved", "peak", "unit", "frac", "achieved_issued", "frac_issued", "avg_launch_ms",
                          "launches_per_step", "traffic", "sustained_clock_ghz")) or None
    if roofline is not None:
        roofline["algorithmic_per_launch"] = rf.get("algorithmic_per_launch")
        roofline["algorithmic_bytes"] = (rf.get("hbm") or {}).get("algorithmic_bytes_per_launch")
        busy = [v for v in (rf.get("mfma_busy") or {}).values() if isinstance(v, (int, float))]
        roofline["mfma_busy_min_max"] = [min(busy), max(busy)] if busy else None      # per instantiation: the detail file
        # achieved / avg_launch_ms: HIP events of THIS run.  traffic / mfma_busy: PMC counters cannot be read from inside the run -
        # they come from the committed profiler pass of the same command; `stale` says whether that pass was made on this build
        roofline["source"] = {"achieved": "HIP events, this run", "traffic": ("committed profile: " + src.split(" ")[0]) if src else None,
                              "stale": rf.get("traffic_stale")}
    ca = _pick(full.get("roofline_cross_attention"), ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_launch_ms", "frac_of_serial_floor"))
    cpu = _pick(full.get("cpu_baseline"), ("value", "unit", "cores", "kind", "host_logical_cpus", "sample"))
    if cpu and isinstance(cpu.get("sample"), str):
        cpu["sample"] = cpu["sample"][:160]
    cfg = _pick(full.get("config"), ("workload", "global_batch", "seq_len", "joints", "embed_dim", "parallelism", "gemm_mode", "streams",
                                     "batches_enqueued_ahead", "lanes"))
    if cfg:
        cfg["workload"] = cfg["workload"][:200]
    others = {}
    for k, v in full.items():
        if not isinstance(v, dict):
            continue
        if k.startswith("variant_"):
            others[k] = v.get("value", "error")
        elif k.startswith("config_"):
            others[k] = next((v[f] for f in ("clips_per_s", "clips_per_s_incl_metrics", "windows_per_s") if f in v), "error")
    lat = full.get("latency") or {}
    rec = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                    "vs_baseline", "dtype", "data")}
    rec.update({"config": cfg, "roofline": roofline, "cpu_baseline": cpu,
                "sustained": _pick(full.get("sustained"), ("value", "seconds", "steps", "ratio_to_value", "clock_ghz_end")),
                "roofline_cross_attention": ca, "outputs_finite": full.get("outputs_finite"),
                "per_rank_clips_s": full.get("per_rank_clips_s"), "per_rank_ms_per_step": full.get("per_rank_ms_per_step"),
                "ref_equiv_tflops": full.get("ref_equiv_tflops"),
                "kernel_ms_total_single_stream": full.get("kernel_ms_total_single_stream"),
                "latency_b1_p50_ms": ((lat.get("B1") or {}).get("eager") or {}).get("p50_ms"),
                "metric_reduction": _pick(full.get("metric_reduction"), ("clips_counted", "backend")),
                "other_records_clips_s": others or None, "library_build_id": full.get("library_build_id"),
                "detail_file": os.path.basename(detail_file) if detail_file else None})
    # hard bound: drop optional fields (last first) until the line fits
    for k in ("other_records_clips_s", "latency_b1_p50_ms", "roofline_cross_attention", "per_rank_ms_per_step", "kernel_ms_total_single_stream",
              "ref_equiv_tflops", "sustained"):
        if len(json.dumps(rec)) <= COMPACT_LIMIT:
            break
        rec.pop(k, None)
    return rec


def emit(full: dict, detail_file: str | None):
    """Full record -> the detail file (and one stderr note); compact record -> stdout, as its only line."""
    if detail_file:
        try:
            with open(detail_file, "w") as fh:
                fh.write(json.dumps(full) + "\n")          # one line: scripts/show_bench.py pretty-prints it
        except OSError as e:
            sys.stderr.write(f"[bench] could not write {detail_file}: {e}\n")
            detail_file = None
    sys.stderr.write(f"[bench] full record ({len(json.dumps(full))} bytes): {detail_file}\n")
    _print_record(json.dumps(compact_record(full, detail_file)))


if __name__ == "__main__":
    main()
