"""GPU parity of the on-device metrics (csrc/metrics.hip) vs the reference's own metric functions (golden) and the oracle."""
import os.path as osp
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, osp.join(osp.dirname(osp.abspath(__file__)), "golden"))
from make_golden_metrics import inputs, smpl_like_regressor  # noqa: E402
from oracle import metrics_oracle as MO  # noqa: E402

pytestmark = pytest.mark.gpu


def test_metrics_match_reference(golden):
    from pmce_amd import assets
    from pmce_amd.eval import Evaluator
    dev = torch.device("cuda:0")
    z = golden("metrics.npz")
    pred, gt, seq = inputs()
    jr = assets.load_j_regressor("h36m").astype(np.float32)
    ev = Evaluator(dev, root_regressor_row=smpl_like_regressor()[0])
    # compute_both_err: same call as Tester.test (mm tensors)
    pm, gm = torch.from_numpy(pred).to(dev), torch.from_numpy(gt).to(dev)
    J = torch.from_numpy(jr).to(dev)
    pj, gj = torch.matmul(J[None], pm), torch.matmul(J[None], gm)
    j_err, s_err = ev.compute_both_err(pm, gm, pj, gj)
    print(f"compute_both_err: joint {j_err:.5f} (ref {float(z['j_err']):.5f}) mesh {s_err:.5f} (ref {float(z['s_err']):.5f})")
    assert abs(j_err - float(z["j_err"])) < 1e-3 and abs(s_err - float(z["s_err"])) < 1e-3          # mm
    # evaluate(): meshes in metres as they leave the model
    mv, mj, pa, pe, ge = ev.per_sample(pm / 1000, gm / 1000)
    e_mv = np.abs(mv.cpu().numpy() - z["mpvpe_mean_per_sample"]).max()
    e_mj = np.abs(mj.cpu().numpy() - z["mpjpe"].mean(1)).max()
    e_pa = np.abs(pa.cpu().numpy() - z["pampjpe"].mean(1)).max()
    print(f"per-sample vs reference: MPVPE {e_mv:.2e} MPJPE {e_mj:.2e} PA-MPJPE {e_pa:.2e} mm")
    assert e_mv < 1e-3 and e_mj < 1e-3 and e_pa < 1e-3
    res = ev.evaluate(pm / 1000, gm / 1000, seq)
    print(res)
    assert abs(res["ACCEL"] * len(seq) - float(z["acc_error_sum"])) < 1e-2
    assert abs(res["MPJPE"] - z["mpjpe"].mean()) < 1e-3 and abs(res["PA-MPJPE"] - z["pampjpe"].mean()) < 1e-3
    assert abs(res["MPVPE"] - z["mpvpe_mean_per_sample"].mean()) < 1e-3


def test_procrustes_edge_cases():
    """reflection branch (det(R) < 0) and an exact similarity, vs the oracle."""
    from pmce_amd.eval import Evaluator
    dev = torch.device("cuda:0")
    ev = Evaluator(dev)
    rng = np.random.default_rng(1)
    B, V = 4, 64
    pj = rng.standard_normal((B, 17, 3)).astype(np.float32) * 100
    gj = pj.copy()
    gj[0] = gj[0] * 1.3 + 5.0                       # exact similarity -> PA error 0
    gj[1][:, 0] *= -1                                # mirrored target -> reflection branch
    gj[2] += rng.standard_normal((17, 3)).astype(np.float32) * 20
    gj[3] = rng.standard_normal((17, 3)).astype(np.float32) * 100
    mesh = np.zeros((B, V, 3), dtype=np.float32)
    t = lambda a: torch.from_numpy(a).to(dev)
    _, mj, pa, _, _ = ev._sample_errors(t(mesh), t(mesh), 1.0, None, None, t(pj), t(gj), None, False)
    idx = list(MO.H36M_EVAL_JOINT)
    for b in range(B):
        Pb = (pj[b] - pj[b][0])[idx].astype(np.float64)
        Gb = (gj[b] - gj[b][0])[idx].astype(np.float64)
        ref_pa = np.sqrt(((MO.rigid_align(Pb, Gb) - Gb) ** 2).sum(1)).mean()
        ref_mj = np.sqrt(((Pb - Gb) ** 2).sum(1)).mean()
        assert abs(float(pa[b]) - ref_pa) < 1e-3 * max(1.0, ref_pa), (b, float(pa[b]), ref_pa)
        assert abs(float(mj[b]) - ref_mj) < 1e-3 * max(1.0, ref_mj)
    assert float(pa[0]) < 1e-3


def test_h36m_flavour_matches_reference(golden):
    """Evaluator with annotated GT joints + camera filter == Human36M.evaluate / compute_both_err (golden from the reference)."""
    from make_golden_metrics_h36m import gt_joints, layout
    from pmce_amd import assets
    from pmce_amd.eval import Evaluator
    dev = torch.device("cuda:0")
    z = golden("metrics_h36m.npz")
    pred, gt, _ = inputs()
    cams, _, seqs = layout(len(pred))
    gj = gt_joints(gt)
    ev = Evaluator(dev, root_regressor_row=smpl_like_regressor()[0])
    pm, gm = torch.from_numpy(pred).to(dev), torch.from_numpy(gt).to(dev)
    J = torch.from_numpy(assets.load_j_regressor("h36m").astype(np.float32)).to(dev)
    j_err, s_err = ev.compute_both_err(pm, gm, torch.matmul(J[None], pm), torch.from_numpy(gj).to(dev))
    assert abs(j_err - float(z["j_err"])) < 1e-3 and abs(s_err - float(z["s_err"])) < 1e-3
    keep = cams == 4
    mv, mj, pa, _, _ = ev.per_sample(pm / 1000, gm / 1000, torch.from_numpy(gj).to(dev))
    e = [np.abs(a.cpu().numpy()[keep] - z[k]).max() for a, k in ((mv, "mpvpe"), (mj, "mpjpe"), (pa, "pampjpe"))]
    print("per-sample vs Human36M.evaluate: MPVPE %.2e MPJPE %.2e PA-MPJPE %.2e mm" % tuple(e))
    assert max(e) < 1e-3
    res = ev.evaluate(pm / 1000, gm / 1000, seqs, gt_joints_mm=torch.from_numpy(gj).to(dev), keep_global=keep)
    print(res)
    assert res["samples"] == int(z["n"])
    assert abs(res["ACCEL"] * res["samples"] - float(z["acc_error_sum"])) < 1e-2
    assert abs(res["MPJPE"] - z["mpjpe"].mean()) < 1e-3 and abs(res["PA-MPJPE"] - z["pampjpe"].mean()) < 1e-3
    assert abs(res["MPVPE"] - z["mpvpe"].mean()) < 1e-3


def test_running_eval_equals_one_shot(golden):
    """RunningEval (batch by batch, meshes dropped) == Evaluator.evaluate on the whole set, for both dataset flavours."""
    from make_golden_metrics_h36m import gt_joints, layout
    from pmce_amd.eval import Evaluator, RunningEval
    dev = torch.device("cuda:0")
    pred, gt, seq = inputs()
    pm, gm = torch.from_numpy(pred).to(dev) / 1000, torch.from_numpy(gt).to(dev) / 1000
    ev = Evaluator(dev, root_regressor_row=smpl_like_regressor()[0])
    want = ev.evaluate(pm, gm, seq)
    run = RunningEval(ev)
    for a, b in ((0, 3), (3, 4), (4, 10)):
        run.add(pm[a:b], gm[a:b])
    assert run.finish(seq) == want
    cams, _, seqs = layout(len(pred))
    gj = torch.from_numpy(gt_joints(gt)).to(dev)
    want = ev.evaluate(pm, gm, seqs, gt_joints_mm=gj, keep_global=cams == 4)
    run = RunningEval(ev)
    for a, b in ((0, 6), (6, 10)):
        run.add(pm[a:b], gm[a:b], gj[a:b])
    assert run.finish(seqs, keep_global=cams == 4) == want


def test_pose_only_flavours_match_reference(golden):
    """Evaluator.compute_joint_err / evaluate_joint (and the MPII3D flavour's compute_both_err / evaluate) on the device against fixtures made
    from the reference's own Human36M / PW3D / MPII3D functions (tests/golden/make_golden_metrics_pose.py), and RunningEval.add_joints batch
    by batch == the one-shot call."""
    from make_golden_metrics_pose import pose_inputs
    from pmce_amd.eval import Evaluator, RunningEval
    dev = torch.device("cuda:0")
    z = golden("metrics_pose.npz")
    t = lambda a: torch.from_numpy(a).to(dev)
    for name, J, seed, key, use_cam in (("pose_h36m", 17, 7, "h36m", True), ("pose_pw3d", 19, 7, "pw3d", False), ("mpii3d", 17, 9, "mpii3d", False)):
        pred, gt, seq, cams = pose_inputs(J, seed=seed)
        ev = Evaluator.for_flavour(name, dev)
        keep = (cams == 4) if use_cam else None
        want_err = float(z["mpii3d_both_joint"]) if name == "mpii3d" else float(z[f"{key}_joint_err"])
        got_err = ev.compute_joint_err(t(pred), t(gt))
        assert abs(got_err - want_err) < 1e-3, (name, got_err, want_err)
        if name == "mpii3d":                                         # MPII3D.compute_both_err: the joints only, mesh error 0
            assert ev.compute_both_err(None, None, t(pred), t(gt)) == (got_err, 0.0)
        mj, pa, pe, ge = ev.joint_errors(t(pred), t(gt))
        sel = keep if keep is not None else np.ones(len(seq), bool)
        e_mj = np.abs(mj.cpu().numpy()[sel] - z[f"{key}_mpjpe"]).max()
        e_pa = np.abs(pa.cpu().numpy()[sel] - z[f"{key}_pampjpe"]).max()
        res = ev.evaluate_joint(t(pred), t(gt), seq, keep_global=keep)
        print(name, f"per-sample vs reference: MPJPE {e_mj:.2e} PA-MPJPE {e_pa:.2e} mm;", res)
        assert e_mj < 1e-3 and e_pa < 1e-3
        assert res["MPVPE"] is None and res["samples"] == int(sel.sum())
        assert abs(res["MPJPE"] - z[f"{key}_mpjpe"].mean()) < 1e-3 and abs(res["PA-MPJPE"] - z[f"{key}_pampjpe"].mean()) < 1e-3
        assert abs(res["ACCEL"] * res["samples"] - float(z[f"{key}_acc_sum"])) < 1e-2
        ref = MO.evaluate_joint_samples(pred, gt, seq, MO.POSE_FLAVOURS[name]["root"], MO.POSE_FLAVOURS[name]["eval_joint"], keep=keep)
        assert abs(res["ACCEL"] - ref["ACCEL"]) < 1e-3 and abs(res["PA-MPJPE"] - ref["PA_MPJPE"]) < 1e-3
        run = RunningEval(ev)
        for a, b in ((0, 5), (5, 6), (6, len(seq))):
            run.add_joints(t(pred[a:b]), t(gt[a:b]))
        assert run.finish(seq, keep_global=keep) == res
    with pytest.raises(ValueError):
        Evaluator.for_flavour("pose_pw3d", dev).compute_joint_err(t(pred), t(gt))      # 17 joints given to the 19-joint COCO flavour
