"""Pin oracle/metrics_oracle.py against the reference's own metric functions (tests/golden/metrics.npz). CPU-only."""
import os.path as osp
import sys

import numpy as np

sys.path.insert(0, osp.join(osp.dirname(osp.abspath(__file__)), "golden"))
from make_golden_metrics import inputs, smpl_like_regressor  # noqa: E402  (input generators only; no reference import)
from oracle import metrics_oracle as MO  # noqa: E402
from pmce_amd import assets  # noqa: E402


def test_metrics_oracle_matches_reference(golden):
    z = golden("metrics.npz")
    pred, gt, seq = inputs()
    assert np.array_equal(seq, z["seq"])
    jr = assets.load_j_regressor("h36m").astype(np.float32)
    pj = np.einsum("jv,nvl->njl", jr, pred)          # fp32 like torch.matmul(J[None], mesh)
    gj = np.einsum("jv,nvl->njl", jr, gt)
    j_err, s_err = MO.compute_both_err(pred, gt, pj, gj)
    assert abs(j_err - float(z["j_err"])) < 2e-3 and abs(s_err - float(z["s_err"])) < 2e-3   # mm; reference runs this in fp32
    ev = list(MO.H36M_EVAL_JOINT)
    A, B = pj[0, ev].astype(np.float64), gj[0, ev].astype(np.float64)
    # the fixture's A/B came from torch fp32 matmul: compare the transforms loosely, the algorithm tightly below
    assert np.abs(MO.rigid_align(A, B) - z["aligned"]).max() < 5e-3
    Ar = A.copy(); Ar[:, 0] *= -1
    assert np.abs(MO.rigid_align(Ar, B) - z["aligned_refl"]).max() < 5e-3
    r = MO.evaluate_samples(pred, gt, smpl_like_regressor(), 0, assets.load_j_regressor("h36m").astype(np.float32), seq)
    assert np.abs(r["mpjpe"] - z["mpjpe"]).max() < 1e-9          # float64 on both sides: identical arithmetic
    assert np.abs(r["pampjpe"] - z["pampjpe"]).max() < 1e-8
    assert np.abs(r["mpvpe"].mean(1) - z["mpvpe_mean_per_sample"]).max() < 1e-9
    assert np.abs(r["mpvpe"][0, :64] - z["mpvpe_first"]).max() < 1e-9
    assert abs(r["ACCEL"] * len(seq) - float(z["acc_error_sum"])) < 1e-8
    printed = str(z["printed"])
    for tag, val in (("H36M MPJPE", r["MPJPE"]), ("PA-MPJPE", r["PA_MPJPE"]), ("MPVPE", r["MPVPE"]), ("ACCEL", r["ACCEL"])):
        line = [l for l in printed.splitlines() if tag in l][0]
        assert abs(float(line.split("tot:")[1]) - val) < 5.1e-3, (tag, line, val)


def test_rigid_align_known_answers():
    rng = np.random.default_rng(0)
    A = rng.standard_normal((14, 3)) * 100
    th = 0.7
    Rz = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
    B = 1.7 * A @ Rz.T + np.array([10.0, -20.0, 5.0])
    assert np.abs(MO.rigid_align(A, B) - B).max() < 1e-9            # exact similarity is recovered
    c, R, t = MO.rigid_transform_3D(A, B)
    assert abs(c - 1.7) < 1e-12 and np.abs(R - Rz).max() < 1e-12
    P = np.cumsum(rng.standard_normal((6, 14, 3)), 0)
    acc = MO.compute_error_accel(joints_gt=P, joints_pred=P + np.arange(6)[:, None, None] ** 2 * 0.5)   # constant accel offset 1.0/axis
    assert np.allclose(acc, np.sqrt(3.0))


def test_h36m_flavour_matches_reference(golden):
    """Human36M.evaluate / compute_both_err (data/Human36M/dataset.py:611-623,715-848): camera-4 filter, annotated GT joints."""
    from make_golden_metrics_h36m import gt_joints, layout
    z = golden("metrics_h36m.npz")
    pred, gt, _ = inputs()
    cams, _, seqs = layout(len(pred))
    gj = gt_joints(gt)
    jr = assets.load_j_regressor("h36m").astype(np.float64)
    pj = np.einsum("jv,nvc->njc", jr, pred.astype(np.float64))
    j_err, s_err = MO.compute_both_err(pred.astype(np.float64), gt.astype(np.float64), pj, gj.astype(np.float64))
    assert abs(j_err - float(z["j_err"])) < 2e-3 and abs(s_err - float(z["s_err"])) < 2e-3     # reference runs this one in fp32
    r = MO.evaluate_samples(pred, gt, smpl_like_regressor(), 0, jr, seqs, gt_joints=gj, keep=cams == 4)
    assert r["mpjpe"].shape[0] == int(z["n"]) == 8
    # the reference multiplies float32 regressors with float32 meshes here; the oracle works in float64
    np.testing.assert_allclose(r["mpjpe"].mean(1), z["mpjpe"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(r["pampjpe"].mean(1), z["pampjpe"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(r["mpvpe"].mean(1), z["mpvpe"], rtol=0, atol=1e-4)
    assert abs(r["ACCEL"] * int(z["n"]) - float(z["acc_error_sum"])) < 1e-3


def test_pose_only_flavours_match_the_reference(golden):
    """The pose-only evaluation flavours of the oracle against fixtures made from the reference's own functions
    (tests/golden/make_golden_metrics_pose.py): Human36M.compute_joint_err / evaluate_joint (camera-4 samples, 14 eval joints), PW3D's (COCO
    set, root = joint [-2], every joint) and MPII3D.compute_both_err / evaluate (17 joints, root 0, every joint; mesh error 0)."""
    sys.path.insert(0, osp.join(osp.dirname(osp.abspath(__file__)), "golden"))
    from make_golden_metrics_pose import pose_inputs
    z = golden("metrics_pose.npz")
    # Human3.6M
    pred, gt, seq, cams = pose_inputs(17)
    f = MO.POSE_FLAVOURS["pose_h36m"]
    assert abs(MO.compute_joint_err(pred, gt, f["root"], f["eval_joint"]) - float(z["h36m_joint_err"])) < 1e-3
    r = MO.evaluate_joint_samples(pred, gt, seq, f["root"], f["eval_joint"], keep=cams == 4)
    assert len(r["mpjpe"]) == int(z["h36m_n"]) == int((cams == 4).sum())
    assert np.abs(r["mpjpe"].mean(1) - z["h36m_mpjpe"]).max() < 1e-4 and np.abs(r["pampjpe"].mean(1) - z["h36m_pampjpe"]).max() < 1e-4
    assert abs(r["acc_sum"] - float(z["h36m_acc_sum"])) < 1e-3
    # 3DPW (COCO joint set)
    pred, gt, seq, _ = pose_inputs(19)
    f = MO.POSE_FLAVOURS["pose_pw3d"]
    assert abs(MO.compute_joint_err(pred, gt, f["root"], f["eval_joint"]) - float(z["pw3d_joint_err"])) < 1e-3
    r = MO.evaluate_joint_samples(pred, gt, seq, f["root"], f["eval_joint"])
    assert np.abs(r["mpjpe"].mean(1) - z["pw3d_mpjpe"]).max() < 1e-4 and np.abs(r["pampjpe"].mean(1) - z["pw3d_pampjpe"]).max() < 1e-4
    assert abs(r["acc_sum"] - float(z["pw3d_acc_sum"])) < 1e-3
    # MPI-INF-3DHP
    pred, gt, seq, _ = pose_inputs(17, seed=9)
    f = MO.POSE_FLAVOURS["mpii3d"]
    assert abs(MO.compute_joint_err(pred, gt, f["root"], f["eval_joint"]) - float(z["mpii3d_both_joint"])) < 1e-3 and float(z["mpii3d_both_mesh"]) == 0.0
    r = MO.evaluate_joint_samples(pred, gt, seq, f["root"], f["eval_joint"])
    assert np.abs(r["mpjpe"].mean(1) - z["mpii3d_mpjpe"]).max() < 1e-4 and np.abs(r["pampjpe"].mean(1) - z["mpii3d_pampjpe"]).max() < 1e-4
    assert abs(r["acc_sum"] - float(z["mpii3d_acc_sum"])) < 1e-3
