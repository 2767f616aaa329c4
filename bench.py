#!/usr/bin/env python3
"""bench.py — PMCE hot-path throughput on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" is one pass of the full two-stream hot path (temporal pose encoder + CoEvoDecoder + 6890-vertex
upsample + J_regressor projection) over one batch of synthetic 16-frame clips per GPU (BASELINE.json configs[2]:
batch = 256, J = 17, C = 256), inputs resident in HBM.  Prints ONE JSON line (rank 0) with the whole-job clips/s,
the roofline of the dominant kernel class (from HIP-event timings taken inside this run), and the oracle's CPU
baseline on the host cores (rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_F32_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: fp32 MFMA == fp32 vector peak
PEAK_HBM_GBS = 8000.0       # HBM3E peak


def gemm_lifter_flops(B, J, C, depth=3, T=16, F=2048):
    M = B * T * J
    return 2.0 * B * T * F * C + depth * 2 * (2.0 * M * C * 3 * C + 2.0 * M * C * C + 2 * 2.0 * M * C * 2 * C)


# kernel function behind each timing class of model.cpp
KERNEL_OF = {"gemm_lifter": "gemm_nt_kernel", "gemm_gru_in": "gemm_nt_kernel", "gemm_ada": "gemm_nt_kernel",
             "gemm_final": "gemm_nt_kernel"}


def class_work(name, B, J, C):
    """ALGORITHMIC work one forward asks of a kernel class (2*M*N*K of the products actually launched; the pruned GRU
    layer-1 steps and the dead joint stream are NOT counted as work): (amount, unit-kind)."""
    GH, F = 1024, 2048
    if name == "gemm_lifter":
        return gemm_lifter_flops(B, J, C), "flop"
    if name == "gemm_gru_in":          # layer 0: 16 steps x 2 directions; layer 1: 9 fwd + 8 bwd steps
        return 2.0 * 16 * B * 6 * GH * F + 2.0 * 17 * B * 3 * GH * 2 * GH, "flop"
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


def pmc_traffic_per_launch(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC pass of this same command
    (profiles/pmc_hbm_traffic_per_launch.json, made by scripts/gpu_pmc.sh: separate FETCH_SIZE / WRITE_SIZE passes, KiB
    units, FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 correction for wide coalesced reads).  bench.py cannot run
    the profiler on itself; without the file the field is null."""
    path = os.path.join(REPO, "profiles", "pmc_hbm_traffic_per_launch.json")
    if not os.path.exists(path):
        return None, None
    try:
        data = json.load(open(path))
        tot_b, tot_n = 0.0, 0
        for name, v in data.items():
            if kernel in name and v.get("launches", 0) > 0:
                tot_b += v["launches"] * (2.0 * v["fetch_kib_raw"] + v["write_kib"]) * 1024.0
                tot_n += v["launches"]
        if tot_n == 0:
            return None, None
        return round(tot_b / tot_n), "profiles/pmc_hbm_traffic_per_launch.json (2*FETCH_SIZE + WRITE_SIZE, KiB, launch-weighted)"
    except Exception:
        return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="clips per GPU per step")
    ap.add_argument("--joints", type=int, default=17)
    ap.add_argument("--embed-dim", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--single-stream", action="store_true",
                    help="keep every launch on one stream (profiling: rocprofv3 then prices each kernel alone)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--pipeline-depth", type=int, default=2,
                    help="batches in flight (models.PMCE.Pipeline): a step's decoder overlaps the next step's pose lifter; "
                         "1 = strictly one batch at a time")
    ap.add_argument("--no-stagger", action="store_true", help="pipeline lanes free-run instead of starting a batch's lifter "
                                                              "when the previous batch's has finished")
    ap.add_argument("--no-host-fed", action="store_true", help="skip the extra (untimed-for-value) host-fed measurement")
    ap.add_argument("--no-variant", action="store_true",
                    help="skip the second timed configuration (the 512-wide pose encoder that BASELINE.json's north_star quotes "
                         "throughput on; every reference config ships 256, which is what `value` is measured on)")
    args = ap.parse_args()

    from pmce_amd import assets, models, sharding, synth

    rank, local, world = sharding.init_from_env()
    assert world == args.gpus or world == 1 and args.gpus == 1, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    if os.environ.get("PMCE_BENCH_SHARE_GPU"):   # plumbing test of the N>1 path on a 1-GPU box (with PMCE_DIST_BACKEND=gloo)
        local = 0
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    B, J, C = args.batch, args.joints, args.embed_dim

    # ---- model (random-init weights of the named architecture; no checkpoints exist offline) ----
    sd = synth.make_state_dict(synth.pmce_spec(J, C, 3), seed=123)
    model = models.PMCE.get_model(J, C, 3)
    model.load_state_dict(sd)
    model.set_j_regressor(assets.load_j_regressor("h36m"))
    model = model.to(dev)

    # ---- synthetic clips, resident in HBM; each rank gets its own shard of the global clip range ----
    lo, hi = sharding.shard_range(B * world, rank, world)
    pose2d_np, feat_np = synth.make_inputs(B, J, seed=1000 + rank)
    pose2d = torch.from_numpy(pose2d_np).to(dev)
    img_feat = torch.from_numpy(feat_np).to(dev)

    depth = 1 if args.single_stream else max(1, args.pipeline_depth)
    pipe = model.pipeline(depth, stagger=not args.no_stagger).prepare(B) if depth > 1 else None
    last = [None]

    def step():
        # one pass of the hot path over one batch; with a pipeline the call returns once the batch is enqueued on its lane
        # (every batch of the timed region is complete before the closing synchronize)
        if pipe is None:
            return model.forward_with_joints(pose2d, img_feat)
        last[0] = pipe.submit(pose2d, img_feat)
        return last[0]

    if args.single_stream:
        model.set_concurrency(False)

    for _ in range(max(args.warmup, 1) if args.warmup > 0 else 0):
        out = step()
    if args.warmup == 0:
        out = step()   # packing / workspace allocation must not be inside the timed region
    torch.cuda.synchronize()
    sharding.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    sharding.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt = sharding.reduce_max(dt, dev)
    clips_per_s = B * world * args.steps / dt

    # final metric reduction — the only collective of the path (RCCL over xGMI): per-rank partial sums
    if pipe is not None:
        out = out.result()
        step = lambda: model.forward_with_joints(pose2d, img_feat)   # the profiling / host-fed passes below run one batch at a time
    mesh, pose, pose3d, pred = out
    partial = torch.stack([pred.abs().sum().double(), mesh.abs().sum().double(),
                           torch.tensor(float(B), device=dev, dtype=torch.float64)])
    total = sharding.reduce_metric_sums(partial)
    finite = bool(torch.isfinite(total).all().item())

    # ---- per-kernel-class timing with HIP events on the launch stream (a few extra, untimed-for-value steps) ----
    model.profile(True)
    nprof = 3
    for _ in range(nprof):
        step()
    torch.cuda.synchronize()
    prof = model.profile_read()
    model.profile(False)
    if args.single_stream:
        model.set_concurrency(False)
    kernel_ms = {k: round(v[0] / nprof, 4) for k, v in prof.items() if v[1] > 0}
    launches = {k: int(v[1] // nprof) for k, v in prof.items() if v[1] > 0}
    # group timing classes by kernel function: the dominant KERNEL is what the roofline prices
    by_kernel = {}
    for k, v in kernel_ms.items():
        by_kernel.setdefault(KERNEL_OF.get(k, k), []).append(k)
    dominant = max(by_kernel, key=lambda kn: sum(kernel_ms[c] for c in by_kernel[kn]))
    dom_classes = by_kernel[dominant]
    dom_ms = sum(kernel_ms[c] for c in dom_classes)
    dom_launches = sum(launches[c] for c in dom_classes)
    works = [class_work(c, B, J, C) for c in dom_classes]
    roofline = None
    if all(w[0] is not None for w in works):
        kind = works[0][1]
        work = sum(w[0] for w in works)
        secs = dom_ms * 1e-3
        traffic, traffic_src = pmc_traffic_per_launch(dominant)
        common = {"kernel": dominant, "classes": dom_classes, "launches_per_step": dom_launches,
                  "avg_launch_ms": round(dom_ms / dom_launches, 5), "traffic": traffic, "traffic_source": traffic_src,
                  "algorithmic_per_launch": work / dom_launches}
        if kind == "flop":
            ach = work / secs / 1e12
            roofline = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(ach / PEAK_F32_TFLOPS, 4), **common}
        else:
            ach = work / secs / 1e9
            roofline = {"bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": round(ach / PEAK_HBM_GBS, 4), **common}
    # the north-star kernel, always reported next to the dominant one
    ca = None
    if "vertex_ca" in kernel_ms:
        w_ca, _ = class_work("vertex_ca", B, J, C)
        ach = w_ca / (kernel_ms["vertex_ca"] * 1e-3) / 1e9
        ca = {"kernel": "vertex_ca", "bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
              "frac": round(ach / PEAK_HBM_GBS, 4), "bytes_per_clip_dir_block": 229376,
              "avg_launch_ms": round(kernel_ms["vertex_ca"] / launches["vertex_ca"], 5)}

    # ---- the same measurement on the 512-wide pose encoder (north_star's "(B,T=16,J=17,C=512)"); all ranks take part ----
    variant = None
    if not args.no_variant and not args.single_stream and C != 512:
        C2 = 512
        sd2 = synth.make_state_dict(synth.pmce_spec(J, C2, 3), seed=123)
        model2 = models.PMCE.get_model(J, C2, 3)
        model2.load_state_dict(sd2)
        model2.set_j_regressor(assets.load_j_regressor("h36m"))
        model2 = model2.to(dev)
        pipe2 = model2.pipeline(depth, stagger=not args.no_stagger).prepare(B) if depth > 1 else None
        run2 = (lambda: pipe2.submit(pose2d, img_feat)) if pipe2 else (lambda: model2.forward_with_joints(pose2d, img_feat))
        k2 = max(5, min(args.steps, 10))
        for _ in range(3):
            o2 = run2()
        torch.cuda.synchronize()
        sharding.barrier()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(k2):
            o2 = run2()
        torch.cuda.synchronize()
        sharding.barrier()
        torch.cuda.synchronize()
        dt2 = sharding.reduce_max(time.perf_counter() - t2, dev)
        o2 = o2.result() if pipe2 else o2
        from pmce_amd.workload import flops_per_clip as _fpc
        v2 = B * world * k2 / dt2
        variant = {"config": {"workload": f"same path, pose encoder width C={C2} (BASELINE.json north_star), batch={B}/GPU, J={J}",
                              "global_batch": B * world, "seq_len": 16, "joints": J, "embed_dim": C2, "batches_in_flight": depth},
                   "value": round(v2, 1), "unit": "clips/s", "steps": k2, "ms_per_step": round(dt2 / k2 * 1e3, 4),
                   "ref_equiv_tflops": round(_fpc(J, C2)["total"] * v2 / 1e12, 2),
                   "outputs_finite": bool(torch.isfinite(o2[0]).all().item())}
        del model2, pipe2, o2, sd2
        torch.cuda.empty_cache()

    # ---- host-fed rate (reported next to `value`, never as it): the same clips start in pageable host memory every step
    # and travel through pmce_amd.staging.PinnedFeeder (memcpy to a pinned ring, async H2D on a copy stream that runs
    # under the previous step's kernels) ----
    host_fed = None
    if rank == 0 and world == 1 and not args.no_host_fed and not args.single_stream:
        from pmce_amd import staging
        nfed = max(5, min(args.steps, 20))
        feeder = staging.PinnedFeeder(dev, {"pose2d": ((B, 16, J, 2), torch.float32), "img_feat": ((B, 16, 2048), torch.float32)}, slots=4)
        host_batches = [{"pose2d": pose2d_np, "img_feat": feat_np}] * (nfed + 2)
        it = feeder.run(host_batches)

        def consume(d):
            if pipe is None:
                model.forward_with_joints(d["pose2d"], d["img_feat"])
            else:                                    # two batches in flight here too: the slot is free when ITS lane is done
                d.release(pipe.submit(d["pose2d"], d["img_feat"]).done)

        for _ in range(2):                       # warm the pinned ring
            consume(next(it))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for d in it:
            consume(d)
        torch.cuda.synchronize()
        t_fed = time.perf_counter() - t1
        host_fed = {"value": round(B * nfed / t_fed, 1), "unit": "clips/s", "steps": nfed,
                    "h2d_bytes_per_step": int(pose2d_np.nbytes + feat_np.nbytes),
                    "path": "pageable host -> pinned ring (4 slots) -> async H2D on a copy stream -> forward (same batches in flight as `value`)"}

    # ---- CPU baseline: the oracle (a port of the reference forward) on this box's host cores ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import pmce_oracle as O
        ncores = os.cpu_count() or 1
        # torch's intra-op pool degrades badly when oversubscribed on many-core hosts: calibrate the thread count
        # on a small batch (a few seconds), then time a bounded sample with the best one.
        p_cpu, f_cpu = (torch.from_numpy(a) for a in synth.make_inputs(64, J, seed=1))
        best_t, best_rate = 1, 0.0
        with torch.no_grad():
            for nt in [t for t in (8, 16, 32, 64, 128) if t <= ncores] or [ncores]:
                torch.set_num_threads(nt)
                O.pmce_forward(sd, p_cpu[:2], f_cpu[:2], model.vj_relation)  # warm the pool
                t1 = time.perf_counter()
                O.pmce_forward(sd, p_cpu[:8], f_cpu[:8], model.vj_relation)
                rate = 8 / (time.perf_counter() - t1)
                if rate > best_rate:
                    best_t, best_rate = nt, rate
            torch.set_num_threads(best_t)
            cb = 64 if best_rate > 8 else 16
            n, t_cpu = 0, 0.0
            while t_cpu < args.cpu_seconds and n < 50:
                t1 = time.perf_counter()
                O.pmce_forward(sd, p_cpu[:cb], f_cpu[:cb], model.vj_relation)
                t_cpu += time.perf_counter() - t1
                n += 1
        cpu = {"value": round(cb * n / t_cpu, 2), "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
               "host_logical_cpus": ncores,
               "sample": f"{n} x batch-{cb} full forwards of oracle/pmce_oracle.py (torch CPU fp32, J={J}, C={C}), "
                         f"thread count calibrated over 8..128"}

    if rank == 0:
        from pmce_amd.workload import flops_per_clip
        flops_clip = flops_per_clip(J, C)["total"]
        line = {
            "metric": "16-frame clips/s", "value": round(clips_per_s, 1), "unit": "clips/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"full two-stream PMCE forward (temporal pose encoder + CoEvoDecoder + 6890-vertex "
                                   f"upsample + J_regressor), batch={B}/GPU, T=16, J={J}, C={C}, random-init weights",
                       "global_batch": B * world, "seq_len": 16, "joints": J, "embed_dim": C,
                       "parallelism": f"clip-sharded dp{world}, weights replicated",
                       "streams": 1 if args.single_stream else 2 * depth, "batches_in_flight": depth},
            "roofline": roofline, "roofline_cross_attention": ca, "cpu_baseline": cpu, "host_fed": host_fed, "variant_c512": variant,
            "kernel_ms_per_step": kernel_ms, "launches_per_step": launches,
            "ref_equiv_tflops": round(flops_clip * clips_per_s / 1e12, 2) if flops_clip else None,
            "outputs_finite": finite,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
