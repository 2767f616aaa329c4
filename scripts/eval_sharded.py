#!/usr/bin/env python3
"""Clip-sharded evaluation harness (BASELINE configs[3]/[4]).  Default: synthetic stand-ins (no datasets are available offline);
with --data-dir the reference's own 3DPW files (data/PW3D/dataset.py:90-128: annotation, ViTPose detections, image features, joint
files) or, with --dataset h36m, its Human3.6M files (data/Human36M/dataset.py:194-269), read by pmce_amd/datasets.py into per-frame tables
that are uploaded ONCE and windowed on the device.

    python scripts/eval_sharded.py --clips 4096 --joints 19                    # 1 GPU, synthetic
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 scripts/eval_sharded.py --clips 35515
    python scripts/eval_sharded.py --data-dir /data/PW3D/pw3d_data [--checkpoint mesh_3dpw.pth.tar]   # real files
    python scripts/eval_sharded.py --dataset h36m --data-dir /data/Human36M/h36m_data [--checkpoint mesh_h36m.pth.tar]
    python scripts/eval_sharded.py --lifter-only [--joints 17|19] [--data-dir ...]     # config/test_pose_{h36m,3dpw}.yml: LiftTester.test
    python scripts/eval_sharded.py --flavour mpii3d                                    # config/test_mesh_mpii3d.yml: joints only, all 17
    python scripts/eval_sharded.py --dataset mpii3d --data-dir /data/MPII3D/mpii3d_data  # the same on the reference's validation files

--lifter-only runs the temporal pose encoder alone (models.PoseEstimation, reference lib/core/base.py:342-387) and evaluates its joints with
``Evaluator.evaluate_joint`` - the Human3.6M flavour (17 joints, root 0, the 14 evaluation joints, camera-4 samples) or the 3DPW flavour (COCO
set of 19, root = Pelvis, every joint); --flavour mpii3d evaluates the joints regressed from the predicted mesh against joint targets the way
MPII3D.evaluate does (data/MPII3D/dataset.py:560-624) - on the synthetic stand-in, or with --dataset mpii3d --data-dir DIR on the reference's
MPI-INF-3DHP validation files (the joblib database + ViTPose output of MPII3D.load_data_val, :249-292; pmce_amd.datasets.load_mpii3d).

Every rank owns a contiguous block of the clip range (weights replicated), runs the HIP forward in batches, computes the
per-sample metrics on the device (pmce_amd.eval) against a synthetic ground truth, and the ranks meet in ONE reduction
(4 floats all_reduce + 14x3 joints all_gather over RCCL).  Prints one JSON line on rank 0.
"""
import argparse, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PMCE_SYNTHETIC_BASE_DATA", "1")   # synthetic weights on the synthetic template (explicit opt-in)


def gt_noise_pool(dev, P=64, seed=4321):
    """Deterministic stand-in ground truth: clip i's mesh = its prediction + 2 cm x pool[(31 i) % P] - a function of the GLOBAL
    clip index only, so every sharding of the set (and a test recomputing it unsharded) sees the same ground truth."""
    g = torch.Generator().manual_seed(seed)
    return (0.02 * torch.randn(P, 6890, 3, generator=g)).to(dev)


def synthetic_gt(mesh, first_index, pool):
    idx = (torch.arange(first_index, first_index + mesh.shape[0], device=mesh.device) * 31) % pool.shape[0]
    return mesh + pool[idx]


def clip_inputs(p_pool, f_pool, first_index, n):
    idx = (torch.arange(first_index, first_index + n, device=p_pool.device) * 7919) % p_pool.shape[0]
    return p_pool[idx], f_pool[idx]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=4096)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--stagger", action="store_true", help="start a batch's lifter when the previous batch's has finished (Pipeline(stagger=True))")
    ap.add_argument("--joints", type=int, default=19)          # 3DPW uses COCO input: J = 19 (PW3D/dataset.py:42,54-55)
    ap.add_argument("--seq-len", type=int, default=500, help="clips per synthetic sequence (for the acceleration error)")
    ap.add_argument("--data-dir", default=None, help="directory holding the reference's 3DPW files (3DPW_latest_<split>.json, ...): "
                                                     "evaluate the real stride-1 window list instead of the synthetic stand-in")
    ap.add_argument("--dataset", default="pw3d", choices=("pw3d", "h36m", "mpii3d"), help="format of --data-dir: the reference's 3DPW files (J = 19) or its "
                                                                              "Human3.6M files (J = 17; the windows of camera 4, as Human36M.evaluate keeps them)")
    ap.add_argument("--split", default="test")
    ap.add_argument("--checkpoint", default=None, help="a reference mesh_*.pth.tar / pose_*.pth.tar (default: deterministic synthetic weights)")
    ap.add_argument("--lifter-only", action="store_true", help="evaluate the pose encoder alone (the reference's test_pose_*.yml / LiftTester)")
    ap.add_argument("--flavour", default=None, choices=("mpii3d",), help="mpii3d: the full model, joints regressed from the mesh, all 17 joints")
    ap.add_argument("--min-seconds", type=float, default=0.0, help="repeat the whole evaluation until this much time has passed; the rate is the "
                                                                   "median pass, the spread is reported (a single pass is 0.7 s at 35 k clips)")
    args = ap.parse_args()
    if args.data_dir and args.dataset == "mpii3d":      # the reference's MPI-INF-3DHP validation files: config/test_mesh_mpii3d.yml
        args.flavour = "mpii3d"
    if args.flavour == "mpii3d":
        if args.lifter_only or (args.data_dir and args.dataset != "mpii3d"):
            ap.error("--flavour mpii3d runs the full model, on the synthetic stand-in or on --dataset mpii3d --data-dir DIR")
        args.joints = 17
    from pmce_amd import models, sharding, synth
    from pmce_amd.eval import Evaluator
    table = win = None
    if args.data_dir:
        from pmce_amd import datasets
        # every rank parses the (host-side) files; the GPU work is sharded
        if args.dataset == "mpii3d":       # (the reference maps its 'test' split to the files' 'val', data/MPII3D/dataset.py:24-25)
            table = datasets.load_mpii3d(args.data_dir, "val" if args.split == "test" else args.split)
        else:
            table = datasets.load_pw3d(args.data_dir, args.split) if args.dataset == "pw3d" else datasets.load_h36m(args.data_dir, args.split)
        win = table.pose_windows(16, 1) if args.lifter_only else table.windows(16, 1)
        if args.dataset == "h36m":                                      # Human36M.evaluate skips every sample whose middle frame is not camera 4
            win = win[table.cam_idxs[datasets.window_mid(win)] == 4]               # (data/Human36M/dataset.py:742-744): they are not run at all here
        args.clips, args.joints = len(win), 17 if args.dataset == "h36m" else 19      # (3DPW and MPI-INF-3DHP feed the COCO set of 19)
    rank, local, world = sharding.init_from_env()
    if os.environ.get("PMCE_BENCH_SHARE_GPU"):      # plumbing runs of the N > 1 path on a box with fewer GPUs (ranks share devices)
        local = local % max(torch.cuda.device_count(), 1)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    J = args.joints
    pose_only = args.lifter_only or args.flavour == "mpii3d"
    want_kind = "lifter" if args.lifter_only else "pmce"
    if args.checkpoint:
        from pmce_amd import checkpoint
        sd, kind, Jc, Cc, depth = checkpoint.load_reference_checkpoint(args.checkpoint)
        if kind == "pmce" and args.lifter_only:                       # a mesh checkpoint holds the lifter under 'pose_lifter.'
            sd, kind = {k[len("pose_lifter."):]: v for k, v in sd.items() if k.startswith("pose_lifter.")}, "lifter"
        assert kind == want_kind and Jc == J, f"{args.checkpoint}: a {kind} checkpoint with J = {Jc}; this run needs a {want_kind} one with J = {J}"
        model = (models.PoseEstimation if args.lifter_only else models.PMCE).get_model(J, Cc, depth)
        model.load_state_dict(sd)
    elif args.lifter_only:
        model = models.PoseEstimation.get_model(J, 256, 3)
        model.load_state_dict(synth.make_state_dict(synth.lifter_spec(J, 256, 3), seed=123))
    else:
        model = models.PMCE.get_model(J, 256, 3)
        model.load_state_dict(synth.make_state_dict(synth.pmce_spec(J, 256, 3), seed=123))
    if args.flavour == "mpii3d":                                       # the caller's J_regressor @ (mesh * 1000) runs inside the forward
        from pmce_amd import assets
        model.set_j_regressor(assets.load_j_regressor("h36m"))
    model = model.to(dev)
    flavour = ("pose_pw3d" if J == 19 else "pose_h36m") if args.lifter_only else (args.flavour or None)
    ev = Evaluator.for_flavour(flavour, dev) if flavour else Evaluator(dev)
    if args.lifter_only and J not in (17, 19):
        raise SystemExit("--lifter-only: the reference evaluates 17 (Human3.6M) or 19 (3DPW, COCO set) joints")
    lo, hi = sharding.shard_range(args.clips, rank, world)
    seq_ids = np.arange(args.clips) // args.seq_len
    gt_mesh = gt_joints = None
    if table is not None:
        from pmce_amd import datasets
        mid = datasets.window_mid(win)                                         # the window's middle frame carries the targets (dataset.py:245-251)
        seq_ids = table.sequence_ids()[mid]
        pose_fr, feat_fr = table.pose2d(dev), table.features_on(dev)    # per-frame tables, uploaded once (8 KB per frame, not 16 x per window)
        if args.lifter_only:    # LiftTester's targets: the COCO-set camera joints (PW3D/dataset.py:241-243) / the annotated joint_cams (Human36M/dataset.py:667)
            tj = table.joints_cam_coco if args.dataset == "pw3d" else table.joints_cam_h36m
            gt_joints = torch.from_numpy(np.ascontiguousarray(tj[mid])).to(dev)
        else:
            gt_joints = torch.from_numpy(table.gt_joints_root_relative()[mid]).to(dev)
            gt_mesh = None if table.gt_mesh_cam is None else table.gt_mesh_cam
    # synthetic inputs for this shard: one pool of `batch` clips, re-indexed (keeps host memory small)
    p_np, f_np = synth.make_inputs(args.batch, J, seed=7)
    p_pool, f_pool = torch.from_numpy(p_np).to(dev), torch.from_numpy(f_np).to(dev)
    pool = gt_noise_pool(dev)
    jpool = pool[:, :J].contiguous() * 1000.0                                  # 20 mm of joint noise for the pose-only stand-ins
    from pmce_amd.eval import RunningEval
    if args.lifter_only:
        model.set_overflow_policy("report")                                    # asynchronous calls; the word is polled once per pass
        pipe = None
    else:
        pipe = model.pipeline(2, stagger=(True if args.stagger else None)).prepare(args.batch)

    def joints_gt(pred_mm, b0):
        if gt_joints is not None:
            return gt_joints[b0:b0 + pred_mm.shape[0]]
        idx = (torch.arange(b0, b0 + pred_mm.shape[0], device=dev) * 31) % jpool.shape[0]
        return pred_mm + jpool[idx]

    def inputs_of(b0, n):
        if table is None:
            return clip_inputs(p_pool, f_pool, b0, n)
        return datasets.window_batch(pose_fr, feat_fr, win[b0:b0 + n])

    def one_pass():
        run = RunningEval(ev)

        def consume(item):
            ticket, b0 = item
            out = ticket.result()
            mesh = out[0]
            if args.flavour == "mpii3d":        # joints regressed from the predicted mesh (base.py:223-225) against joint targets, all 17
                run.add_joints(out[3], joints_gt(out[3], b0))
            elif table is None:
                run.add(mesh, synthetic_gt(mesh, b0, pool))                         # per-sample errors + 14x3 joints; the mesh is dropped
            else:   # annotated joints (mm, root-relative); the mesh target only if the caller supplied SMPL meshes (else MPVPE is void)
                n = mesh.shape[0]
                gm = mesh if gt_mesh is None else torch.from_numpy(np.ascontiguousarray(gt_mesh[mid[b0:b0 + n]])).to(dev) / 1000.0
                run.add(mesh, gm, gt_joints[b0:b0 + n])

        pending = None
        for b0 in range(lo, hi, args.batch):
            n = min(args.batch, hi - b0)
            if args.lifter_only:
                pose3d = model(*inputs_of(b0, n))                                   # [n, J, 3] mm (PoseEstimation.py:95-115)
                run.add_joints(pose3d, joints_gt(pose3d, b0))
                continue
            ticket = pipe.submit(*inputs_of(b0, n), want_joints=args.flavour == "mpii3d")   # two batches in flight
            if pending is not None:
                consume(pending)
            pending = (ticket, b0)
        if pending is not None:
            consume(pending)
        # drains the lanes and polls the model's overflow word BEFORE the metrics are finished (a warning names the batches; every batch was already
        # reduced as it was produced, and non-finite predictions are counted and named per clip in the result - nothing is re-run behind the metrics' back)
        if pipe is not None:
            named = pipe.synchronize()
        else:
            named = ["(some batch)"] if model.overflowed() else []
            model.clear_overflow()
        return run.finish(seq_ids, lo, hi), named

    # a pass over 35 k clips is 0.7 s: with --min-seconds the whole evaluation is repeated and the rate is the MEDIAN pass (spread reported)
    pass_s, res, named = [], None, []
    t_all = time.perf_counter()
    while True:
        torch.cuda.synchronize(); sharding.barrier(); t0 = time.perf_counter()
        res, named = one_pass()
        torch.cuda.synchronize(); sharding.barrier()
        pass_s.append(sharding.reduce_max(time.perf_counter() - t0, dev))
        if sharding.reduce_max(time.perf_counter() - t_all, dev) >= args.min_seconds and (args.min_seconds <= 0 or len(pass_s) >= 3):
            break
    dt = float(np.median(pass_s))
    if rank == 0:
        # where a batch of this configuration spends its time, and the roofline of its dominant kernel (one stream, HIP events)
        import bench
        nb = min(args.batch, args.clips)
        eng = model._ensure_packed()
        eng.set_concurrency(False); eng.profile(True)
        for _ in range(3):
            model(*inputs_of(0, nb))
        torch.cuda.synchronize()
        prof = eng.profile_read()
        eng.profile(False); eng.set_concurrency(True)
        kernel_ms = {k: round(v[0] / 3, 4) for k, v in prof.items() if v[1] > 0}
        launches = {k: int(v[1] // 3) for k, v in prof.items() if v[1] > 0}
        what = ("pose encoder only (LiftTester.test, lib/core/base.py:342-387)" if args.lifter_only else
                "full model, joints regressed from the mesh, MPII3D.evaluate" if args.flavour == "mpii3d" else "full model")
        res.update({"clips": args.clips, "n_gpus": world, "clips_per_s_incl_metrics": round(args.clips / dt, 1), "seconds": round(dt, 4),
                    "passes": len(pass_s), "pass_seconds_min_max": [round(min(pass_s), 4), round(max(pass_s), 4)],
                    "clips_per_s_spread": round((max(pass_s) - min(pass_s)) / dt, 4), "flavour": ev.flavour or f"mesh_{args.dataset}", "model": what,
                    "J": J, "batch": args.batch, "gemm_mode": model.gemm_mode(), "batches_rerun_on_fp32_pipe": named,
                    "metric_reduction": {"collective": "all_reduce(SUM) of 4 fp64 partials + all_gather of 2 x n_eval x 3 joints per clip",
                                         "backend": (torch.distributed.get_backend() if world > 1 else None)},
                    "roofline": bench.dominant_kernel_roofline(kernel_ms, launches, nb, J, 256, model.gemm_mode()),
                    "data": ("synthetic stand-in (no 3DPW / H36M / MPI-INF-3DHP files offline)" if table is None else
                             f"{table.name}: {len(table)} frames, {len(win)} stride-1 windows from {args.data_dir}; targets = "
                             + ("the annotated camera-space joints" if args.lifter_only else "annotated h36m joints"
                                + ("" if gt_mesh is not None else "; no ground-truth meshes supplied: MPVPE is void")))})
        if table is not None and gt_mesh is None:
            res["MPVPE"] = None
        print(json.dumps(res))
    if world > 1:
        sharding.barrier()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
