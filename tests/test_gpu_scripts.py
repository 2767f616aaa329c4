"""The harnesses behind BASELINE.json's other configurations are part of the suite: what bench.py reports as
`config_decoder_b64`, `config_eval_sharded_j19` and `config_streaming` comes from these very scripts.

* scripts/eval_sharded.py with TWO ranks (gloo above the same pmce_amd.sharding code: RCCL refuses two ranks on one device),
  600 clips, J = 19, the rank boundary inside a sequence: its MPJPE / PA-MPJPE / MPVPE / ACCEL against oracle/metrics_oracle.py
  (the reference's evaluate arithmetic, data/PW3D/dataset.py:351-462) on predictions recomputed unsharded in this process;
* scripts/stream_bench.py and scripts/decoder_bench.py on small sizes: they run, count what they should and carry a roofline.
"""
import json
import os
import os.path as osp
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = osp.dirname(osp.abspath(__file__))
REPO = osp.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_script(script, argv, world=1, timeout=900):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, PMCE_SYNTHETIC_BASE_DATA="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        if world > 1:
            env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       PMCE_DIST_BACKEND="gloo", PMCE_BENCH_SHARE_GPU="1")
        else:
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
                env.pop(k, None)
        procs.append(subprocess.Popen([sys.executable, osp.join(REPO, "scripts", script), *argv], env=env, cwd=REPO,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=timeout) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-3000:]
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert lines, outs[0][1][-2000:]
    return json.loads(lines[-1])


def test_eval_sharded_script_two_ranks_vs_metrics_oracle():
    from oracle import metrics_oracle as MO
    from pmce_amd import assets, models, synth
    sys.path.insert(0, osp.join(REPO, "scripts"))
    import eval_sharded as ES
    clips, J, batch, seq_len = 600, 19, 128, 250          # 2 ranks x 300 clips: the boundary cuts sequence 1 (clips 250..499)
    got = _run_script("eval_sharded.py", ["--clips", str(clips), "--joints", str(J), "--batch", str(batch), "--seq-len", str(seq_len)], world=2)
    assert got["n_gpus"] == 2 and got["samples"] == clips and got["nonfinite_samples"] == 0
    assert got["metric_reduction"]["backend"] == "gloo" and got["batches_rerun_on_fp32_pipe"] == []
    assert got["roofline"] and got["roofline"]["frac"] > 0

    # the same clips unsharded, in this process: predictions from the model, ground truth by the script's own rule
    assets.allow_synthetic_base_data()
    dev = torch.device("cuda:0")
    model = models.PMCE.get_model(J, 256, 3)
    model.load_state_dict(synth.make_state_dict(synth.pmce_spec(J, 256, 3), seed=123))
    model = model.to(dev)
    p_np, f_np = synth.make_inputs(batch, J, seed=7)
    p_pool, f_pool = torch.from_numpy(p_np).to(dev), torch.from_numpy(f_np).to(dev)
    pool = ES.gt_noise_pool(dev)
    pred, gt = [], []
    for b0 in range(0, clips, 100):                        # (other batch boundaries than either rank's: clips are independent)
        mesh = model(*ES.clip_inputs(p_pool, f_pool, b0, min(100, clips - b0)))[0]
        pred.append(mesh.double().cpu().numpy())
        gt.append(ES.synthetic_gt(mesh, b0, pool).double().cpu().numpy())
    pred, gt = np.concatenate(pred) * 1000, np.concatenate(gt) * 1000
    jr = assets.load_j_regressor("h36m").astype(np.float32)
    ref = MO.evaluate_samples(pred, gt, jr[:1], 0, jr, np.arange(clips) // seq_len)
    print("2 ranks:", {k: got[k] for k in ("MPVPE", "MPJPE", "PA-MPJPE", "ACCEL")}, "\noracle :", {k: ref[k] for k in ("MPVPE", "MPJPE", "PA_MPJPE", "ACCEL")})
    for k, r in (("MPVPE", "MPVPE"), ("MPJPE", "MPJPE"), ("PA-MPJPE", "PA_MPJPE"), ("ACCEL", "ACCEL")):
        assert abs(got[k] - ref[r]) < 2e-3, (k, got[k], ref[r])          # millimetres


def test_eval_sharded_script_on_reference_format_files(tmp_path):
    """`eval_sharded.py --data-dir` on a directory in the reference's 3DPW file formats (tests/golden/pw3d_files.py; the
    reader itself is pinned against the reference's own loader in tests/test_datasets_host.py): two ranks, the real stride-1 window list
    (67 windows of 4 videos), per-frame tables uploaded once and windowed on the device, annotated h36m joints as targets - against the
    metrics oracle on predictions recomputed unsharded here from host-assembled windows."""
    from oracle import metrics_oracle as MO
    from oracle import staging_oracle as S
    from pmce_amd import assets, datasets, models, synth
    sys.path.insert(0, osp.join(HERE, "golden"))
    import pw3d_files
    path = pw3d_files.write(str(tmp_path))
    got = _run_script("eval_sharded.py", ["--data-dir", path, "--batch", "32"], world=2)
    table = datasets.load_pw3d(path)
    win = table.windows()
    assert got["n_gpus"] == 2 and got["samples"] == len(win) == 67 and got["J"] == 19 and got["MPVPE"] is None and "3DPW test" in got["data"]
    # unsharded, inputs assembled on the HOST with the staging oracle
    assets.allow_synthetic_base_data()
    dev = torch.device("cuda:0")
    model = models.PMCE.get_model(19, 256, 3)
    model.load_state_dict(synth.make_state_dict(synth.pmce_spec(19, 256, 3), seed=123))
    model = model.to(dev)
    ext = np.stack([S.add_pelvis_and_neck(k) for k in table.keypoints])
    pose2d = np.stack([np.asarray(S.normalize_screen_coordinates(ext[i][:, :2], w=table.img_shapes[i][1], h=table.img_shapes[i][0]), dtype=np.float32)
                       for i in range(len(table))])
    fr = datasets.window_frames(win)
    mesh = model(torch.from_numpy(pose2d[fr]).to(dev), torch.from_numpy(table.features[fr]).to(dev))[0]
    pred = mesh.double().cpu().numpy() * 1000
    jr = assets.load_j_regressor("h36m").astype(np.float32)
    mid = win[:, 0] + 8
    ref = MO.evaluate_samples(pred, pred, jr[:1], 0, jr, table.sequence_ids()[mid], gt_joints=table.gt_joints_root_relative()[mid].astype(np.float64))
    print("2 ranks:", {k: got[k] for k in ("MPJPE", "PA-MPJPE", "ACCEL")}, "\noracle :", {k: ref[k] for k in ("MPJPE", "PA_MPJPE", "ACCEL")})
    for k, r in (("MPJPE", "MPJPE"), ("PA-MPJPE", "PA_MPJPE"), ("ACCEL", "ACCEL")):
        assert abs(got[k] - ref[r]) < 2e-3 * max(1.0, abs(ref[r]) / 100), (k, got[k], ref[r])          # millimetres


def test_eval_sharded_script_on_reference_format_h36m_files(tmp_path):
    """Round 5: `eval_sharded.py --dataset h36m --data-dir` on a directory in the reference's Human3.6M file formats (tests/golden/h36m_files.py;
    the reader is pinned against the reference's own Human36M class in tests/test_datasets_host.py): two ranks, J = 17, the stride-1 windows
    whose middle frame is camera 4 (what Human36M.evaluate keeps) - against the metrics oracle on predictions recomputed unsharded here from
    host-assembled windows."""
    from oracle import metrics_oracle as MO
    from oracle import staging_oracle as S
    from pmce_amd import assets, datasets, models, synth
    sys.path.insert(0, osp.join(HERE, "golden"))
    import h36m_files
    path = h36m_files.write(str(tmp_path))
    got = _run_script("eval_sharded.py", ["--dataset", "h36m", "--data-dir", path, "--batch", "16"], world=2)
    table = datasets.load_h36m(path)
    win = table.windows()
    win = win[table.cam_idxs[win[:, 0] + 8] == 4]
    assert got["n_gpus"] == 2 and got["samples"] == len(win) and 0 < len(win) < 76 and got["J"] == 17 and got["MPVPE"] is None and "Human3.6M test" in got["data"]
    assets.allow_synthetic_base_data()
    dev = torch.device("cuda:0")
    model = models.PMCE.get_model(17, 256, 3)
    model.load_state_dict(synth.make_state_dict(synth.pmce_spec(17, 256, 3), seed=123))
    model = model.to(dev)
    pose2d = np.stack([np.asarray(S.normalize_screen_coordinates(table.keypoints[i][:, :2], w=table.img_shapes[i][1], h=table.img_shapes[i][0]), dtype=np.float32)
                       for i in range(len(table))])
    fr = datasets.window_frames(win)
    mesh = model(torch.from_numpy(pose2d[fr]).to(dev), torch.from_numpy(table.features[fr]).to(dev))[0]
    pred = mesh.double().cpu().numpy() * 1000
    jr = assets.load_j_regressor("h36m").astype(np.float32)
    mid = win[:, 0] + 8
    ref = MO.evaluate_samples(pred, pred, jr[:1], 0, jr, table.sequence_ids()[mid], gt_joints=table.gt_joints_root_relative()[mid].astype(np.float64))
    print("2 ranks:", {k: got[k] for k in ("MPJPE", "PA-MPJPE", "ACCEL")}, "\noracle :", {k: ref[k] for k in ("MPJPE", "PA_MPJPE", "ACCEL")})
    for k, r in (("MPJPE", "MPJPE"), ("PA-MPJPE", "PA_MPJPE"), ("ACCEL", "ACCEL")):
        assert abs(got[k] - ref[r]) < 2e-3 * max(1.0, abs(ref[r]) / 100), (k, got[k], ref[r])          # millimetres


def test_stream_bench_script():
    from pmce_amd import streaming
    L = 700
    got = _run_script("stream_bench.py", ["--frames", str(L), "--batch", "128", "--min-seconds", "1"])
    nwin = len(streaming.window_indices(L))      # stride-1 windows, tail dropped to the last full VIBE chunk (lib/_img_utils.py:27-55): 673
    assert nwin == (L // 16) * 16 - 15
    assert got["frames"] == L and got["windows"] == nwin and got["samples"] == nwin
    assert got["windows_per_s"] > 0 and got["nonfinite_samples"] == 0
    # a record, not a glimpse: at least three passes over >= --min-seconds, the median reported with its spread and the whole run's rate
    assert got["passes"] >= 3 and got["sustained"]["seconds"] >= 1.0 and got["sustained"]["windows_per_s"] > 0
    assert got["windows_per_s_min_max"][0] <= got["windows_per_s"] <= got["windows_per_s_min_max"][1]
    assert all(np.isfinite(got[k]) for k in ("MPVPE", "MPJPE", "PA-MPJPE", "ACCEL"))
    r = got["roofline"]
    assert r and r["kernel"] and 0 < r["frac"] < 1 and "gemm_lifter" in got["kernel_ms_per_window_batch"]


def test_eval_sharded_script_on_mpii3d_format_files(tmp_path):
    """`eval_sharded.py --dataset mpii3d --data-dir` on a directory in the reference's MPI-INF-3DHP validation file formats
    (tests/golden/mpii3d_files.py; the reader is pinned against the reference's own ``MPII3D`` class in tests/test_datasets_host.py): two ranks,
    the stride-1 window list of split_into_chunks_pose (50 windows of 2 videos - the third is shorter than a clip), COCO-19 input, the joints
    regressed from the predicted mesh against the annotated joints, all 17, root 0 (config/test_mesh_mpii3d.yml, MPII3D.evaluate) - against
    the metrics oracle on predictions recomputed unsharded here from host-assembled windows."""
    from oracle import metrics_oracle as MO
    from oracle import staging_oracle as S
    from pmce_amd import assets, datasets, models, synth
    sys.path.insert(0, osp.join(HERE, "golden"))
    import mpii3d_files
    path = mpii3d_files.write(str(tmp_path))
    got = _run_script("eval_sharded.py", ["--dataset", "mpii3d", "--data-dir", path, "--batch", "16"], world=2)
    table = datasets.load_mpii3d(path)
    win = table.windows()
    assert got["n_gpus"] == 2 and got["samples"] == len(win) == 50 and got["J"] == 19 and got["flavour"] == "mpii3d"
    assert got["MPVPE"] is None and "MPI-INF-3DHP val" in got["data"]
    assets.allow_synthetic_base_data()
    dev = torch.device("cuda:0")
    model = models.PMCE.get_model(19, 256, 3)
    model.load_state_dict(synth.make_state_dict(synth.pmce_spec(19, 256, 3), seed=123))
    model.set_j_regressor(assets.load_j_regressor("h36m"))
    model = model.to(dev)
    ext = np.stack([S.add_pelvis_and_neck(k) for k in table.keypoints])
    pose2d = np.stack([np.asarray(S.normalize_screen_coordinates(ext[i][:, :2], w=2048, h=2048), dtype=np.float32) for i in range(len(table))])
    fr = datasets.window_frames(win)
    pj = model.forward_with_joints(torch.from_numpy(pose2d[fr]).to(dev), torch.from_numpy(table.features[fr]).to(dev))[3]
    mid = datasets.window_mid(win)
    ref = MO.evaluate_joint_samples(pj.double().cpu().numpy(), table.gt_joints_root_relative()[mid].astype(np.float64), table.sequence_ids()[mid], 0, None)
    print("2 ranks:", {k: got[k] for k in ("MPJPE", "PA-MPJPE", "ACCEL")}, "\noracle :", {k: ref[k] for k in ("MPJPE", "PA_MPJPE", "ACCEL")})
    for k, r in (("MPJPE", "MPJPE"), ("PA-MPJPE", "PA_MPJPE"), ("ACCEL", "ACCEL")):
        assert abs(got[k] - ref[r]) < 2e-3 * max(1.0, abs(ref[r]) / 100), (k, got[k], ref[r])


def test_soak_script_is_deterministic_across_lanes_and_batch_sizes():
    """scripts/soak.py, a few seconds of it: forwards of changing batch sizes through the two pipeline lanes and through direct calls - every
    result bit-identical to the first one of its batch size, clip 0's mesh the same bits at every batch size (the round's 90-second run:
    22,731 forwards, 2.0 M clips, profiles/r06_y_soak.txt)."""
    got = _run_script("soak.py", ["--seconds", "6", "--embed-dim", "256", "--batches", "1,3,16,64,200"])
    assert got["forwards"] > 50 and len(got["per_batch"]) == 5
    assert got["bit_mismatches"] == 0 and got["nonfinite_results"] == 0 and not got["overflow_word"]
    assert got["mesh_of_clip_0_max_abs_across_batch_sizes_m"] == 0.0


def test_decoder_bench_script():
    got = _run_script("decoder_bench.py", ["--batch", "64", "--steps", "10", "--min-seconds", "1"])
    assert got["passes"] >= 3 and got["sustained"]["seconds"] >= 1.0 and got["clips_per_s_min_max"][0] <= got["clips_per_s"] <= got["clips_per_s_min_max"][1]
    assert got["clips_per_s"] > 0 and got["outputs_finite"] and got["cross_attention"]["kernel"] in ("vertex_ca_mlp", "vertex_ca")
    assert got["roofline"] and got["roofline"]["kernel"] and 0 < got["roofline"]["frac"] < 1


def test_eval_sharded_lifter_only_two_ranks_vs_metrics_oracle():
    """config/test_pose_3dpw.yml / test_pose_h36m.yml (LiftTester.test, lib/core/base.py:342-387): the pose encoder alone, evaluated with the
    pose-only flavours - two ranks with the shard boundary inside a sequence against oracle/metrics_oracle.evaluate_joint_samples on the same
    predictions, for the COCO-19 (3DPW) and the 17-joint (Human3.6M) form; repeated passes (--min-seconds) give the same metrics."""
    from oracle import metrics_oracle as MO
    from pmce_amd import models, synth
    sys.path.insert(0, osp.join(REPO, "scripts"))
    import eval_sharded as ES
    dev = torch.device("cuda:0")
    for J, name in ((19, "pose_pw3d"), (17, "pose_h36m")):
        clips, batch, seq_len = 500, 96, 200
        got = _run_script("eval_sharded.py", ["--lifter-only", "--clips", str(clips), "--joints", str(J), "--batch", str(batch), "--seq-len", str(seq_len),
                                              "--min-seconds", "1"], world=2)
        assert got["flavour"] == name and got["n_gpus"] == 2 and got["samples"] == clips and got["MPVPE"] is None and got["passes"] >= 3
        assert got["nonfinite_samples"] == 0 and got["roofline"] and got["roofline"]["frac"] > 0
        model = models.PoseEstimation.get_model(J, 256, 3)
        model.load_state_dict(synth.make_state_dict(synth.lifter_spec(J, 256, 3), seed=123))
        model = model.to(dev)
        p_np, f_np = synth.make_inputs(batch, J, seed=7)
        p_pool, f_pool = torch.from_numpy(p_np).to(dev), torch.from_numpy(f_np).to(dev)
        jpool = ES.gt_noise_pool(dev)[:, :J].contiguous() * 1000.0
        pred, gt = [], []
        for b0 in range(0, clips, 100):
            p3 = model(*ES.clip_inputs(p_pool, f_pool, b0, min(100, clips - b0)))
            idx = (torch.arange(b0, b0 + p3.shape[0], device=dev) * 31) % jpool.shape[0]
            pred.append(p3.double().cpu().numpy()); gt.append((p3 + jpool[idx]).double().cpu().numpy())
        f = MO.POSE_FLAVOURS[name]
        ref = MO.evaluate_joint_samples(np.concatenate(pred), np.concatenate(gt), np.arange(clips) // seq_len, f["root"], f["eval_joint"])
        print(name, {k: got[k] for k in ("MPJPE", "PA-MPJPE", "ACCEL")}, "oracle", {k: ref[k] for k in ("MPJPE", "PA_MPJPE", "ACCEL")})
        for k, r in (("MPJPE", "MPJPE"), ("PA-MPJPE", "PA_MPJPE"), ("ACCEL", "ACCEL")):
            assert abs(got[k] - ref[r]) < 2e-3, (name, k, got[k], ref[r])          # millimetres


def test_eval_sharded_mpii3d_flavour_vs_metrics_oracle():
    """config/test_mesh_mpii3d.yml: the full model, joints regressed from the predicted mesh (base.py:223-225) against joint targets, every one
    of the 17 joints, root 0 (MPII3D.evaluate, data/MPII3D/dataset.py:560-624) - against the metrics oracle on the same predictions."""
    from oracle import metrics_oracle as MO
    from pmce_amd import assets, models, synth
    sys.path.insert(0, osp.join(REPO, "scripts"))
    import eval_sharded as ES
    dev = torch.device("cuda:0")
    clips, batch, seq_len, J = 300, 64, 120, 17
    got = _run_script("eval_sharded.py", ["--flavour", "mpii3d", "--clips", str(clips), "--batch", str(batch), "--seq-len", str(seq_len)], world=2)
    assert got["flavour"] == "mpii3d" and got["samples"] == clips and got["MPVPE"] is None and got["J"] == 17
    assets.allow_synthetic_base_data()
    model = models.PMCE.get_model(J, 256, 3)
    model.load_state_dict(synth.make_state_dict(synth.pmce_spec(J, 256, 3), seed=123))
    model.set_j_regressor(assets.load_j_regressor("h36m"))
    model = model.to(dev)
    p_np, f_np = synth.make_inputs(batch, J, seed=7)
    p_pool, f_pool = torch.from_numpy(p_np).to(dev), torch.from_numpy(f_np).to(dev)
    jpool = ES.gt_noise_pool(dev)[:, :J].contiguous() * 1000.0
    pred, gt = [], []
    for b0 in range(0, clips, 100):
        pj = model.forward_with_joints(*ES.clip_inputs(p_pool, f_pool, b0, min(100, clips - b0)))[3]
        idx = (torch.arange(b0, b0 + pj.shape[0], device=dev) * 31) % jpool.shape[0]
        pred.append(pj.double().cpu().numpy()); gt.append((pj + jpool[idx]).double().cpu().numpy())
    ref = MO.evaluate_joint_samples(np.concatenate(pred), np.concatenate(gt), np.arange(clips) // seq_len, 0, None)
    print({k: got[k] for k in ("MPJPE", "PA-MPJPE", "ACCEL")}, "oracle", {k: ref[k] for k in ("MPJPE", "PA_MPJPE", "ACCEL")})
    for k, r in (("MPJPE", "MPJPE"), ("PA-MPJPE", "PA_MPJPE"), ("ACCEL", "ACCEL")):
        assert abs(got[k] - ref[r]) < 2e-3, (k, got[k], ref[r])
