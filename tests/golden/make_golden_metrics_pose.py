#!/usr/bin/env python3
"""Golden vectors for the POSE-ONLY evaluation flavours from the REAL reference functions (build container only):

* ``Human36M.compute_joint_err`` / ``Human36M.evaluate_joint`` (data/Human36M/dataset.py:600-609,625-713): root 0, the 14 eval joints, camera-4
  samples only, targets = the annotated ``joint_cams`` of the window's middle frame;
* ``PW3D.compute_joint_err`` / ``PW3D.evaluate_joint`` (data/PW3D/dataset.py:260-267,284-349): COCO joint set (19), root = joint [-2], every joint;
* ``MPII3D.compute_joint_err`` / ``MPII3D.compute_both_err`` / ``MPII3D.evaluate`` (data/MPII3D/dataset.py:539-624): 17 joints, root 0, every
  joint; the mesh error is reported as 0.

All called as unbound methods on a stub ``self`` (the dataset classes cannot be constructed offline); the evaluate functions print 2-decimal
totals only, so their per-sample arrays are captured from the frame locals with a profile hook.  Outputs only; the tests regenerate the inputs
from pmce_amd.synth (``pose_inputs`` below is imported by them)."""
import contextlib, io, os.path as osp, sys, types
import numpy as np
import torch

HERE = osp.dirname(osp.abspath(__file__)); REPO = osp.dirname(osp.dirname(HERE)); REF = "/root/reference"
sys.path.insert(0, REPO); sys.path.insert(0, HERE)
from pmce_amd import synth  # noqa: E402

EVAL14 = (1, 2, 3, 4, 5, 6, 8, 10, 11, 12, 13, 14, 15, 16)


def pose_inputs(J, N=12, seed=7):
    """Predicted / target joints in mm ([N,J,3]): a smooth skeleton-sized base, a per-sample offset (removed by the root alignment) and noise
    of a few cm; sequences 0..2 of 5 / 4 / 3 samples; samples 3 and 8 are 'not camera 4'."""
    u = synth.uniform_pm1
    base = u(f"pose.base.{J}", N * J * 3, seed).reshape(N, J, 3) * np.array([400.0, 800.0, 200.0], dtype=np.float32)
    gt = base + u(f"pose.gtoff.{J}", N * 3, seed).reshape(N, 1, 3) * 2500.0
    pred = base * 1.04 + u(f"pose.noise.{J}", N * J * 3, seed).reshape(N, J, 3) * 45.0 + u(f"pose.poff.{J}", N * 3, seed).reshape(N, 1, 3) * 300.0
    seq = np.array([0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2][:N])
    cams = np.array([4, 4, 4, 2, 4, 4, 4, 4, 1, 4, 4, 4][:N])
    return pred.astype(np.float32), gt.astype(np.float32), seq, cams


def capture(fn_name, keys, call):
    cap = {}

    def prof(frame, event, arg):
        if event == "return" and frame.f_code.co_name == fn_name:
            for k in keys:
                cap[k] = np.array(frame.f_locals[k], dtype=np.float64)
    sys.setprofile(prof)
    with contextlib.redirect_stdout(io.StringIO()) as buf:
        call()
    sys.setprofile(None)
    cap["printed"] = buf.getvalue()
    return cap


def main():
    from make_golden_metrics_h36m import shims
    shims()
    aug = sys.modules["aug_utils"]                      # (MPII3D's module imports two more names of the absent-cv2 aug_utils)
    aug.j2d_processing = aug.transform_joint_to_other_db = None
    from Human36M.dataset import Human36M
    from PW3D.dataset import PW3D
    from MPII3D.dataset import MPII3D
    out = {}
    # ---- Human3.6M pose-only ----
    pred, gt, seq, cams = pose_inputs(17)
    N = len(seq)
    st = types.SimpleNamespace(human36_eval_joint=EVAL14, vid_indices=[(n, n) for n in range(N)], seqlen=16, cam_idxs=cams, joint_cams=gt,
                               img_names=[f"seq{seq[n]}_{n:06d}.jpg" for n in range(N)])
    out["h36m_joint_err"] = float(Human36M.compute_joint_err(st, torch.from_numpy(pred), torch.from_numpy(gt)))
    outs = [dict(joint_coord=pred[n], joint_coord_target=gt[n]) for n in range(N)]
    c = capture("evaluate_joint", ("mpjpe", "pampjpe", "acc_error_h36m", "i"), lambda: Human36M.evaluate_joint(st, outs))
    out.update(h36m_mpjpe=c["mpjpe"].mean(1), h36m_pampjpe=c["pampjpe"].mean(1), h36m_acc_sum=float(c["acc_error_h36m"]), h36m_n=int(c["i"]),
               h36m_printed=c["printed"])
    print(c["printed"])
    # ---- 3DPW pose-only (COCO joint set, 19 joints, root = [-2]) ----
    pred, gt, seq, _ = pose_inputs(19)
    st = types.SimpleNamespace(coco_joint_num=19, vid_indices=[(n, n) for n in range(N)], seqlen=16, vid_names=[f"seq{seq[n]}" for n in range(N)])
    out["pw3d_joint_err"] = float(PW3D.compute_joint_err(st, torch.from_numpy(pred), torch.from_numpy(gt)))
    outs = [dict(joint_coord=pred[n], joint_coord_target=gt[n]) for n in range(N)]
    c = capture("evaluate_joint", ("mpjpe", "pa_mpjpe", "acc_error_h36m"), lambda: PW3D.evaluate_joint(st, outs))
    out.update(pw3d_mpjpe=c["mpjpe"].mean(1), pw3d_pampjpe=c["pa_mpjpe"].mean(1), pw3d_acc_sum=float(c["acc_error_h36m"]), pw3d_printed=c["printed"])
    print(c["printed"])
    # ---- MPI-INF-3DHP (config/test_mesh_mpii3d.yml): 17 joints, root 0, every joint ----
    pred, gt, seq, _ = pose_inputs(17, seed=9)
    st = types.SimpleNamespace(human36_joint_num=17, vid_indices=[(n, n) for n in range(N)], seqlen=16,
                               img_paths=[f"S1/Seq{seq[n]}/img_{n:06d}.jpg" for n in range(N)])          # [:-11] = the sequence
    out["mpii3d_joint_err"] = float(MPII3D.compute_joint_err(st, torch.from_numpy(np.concatenate([pred, pred[:, :2]], 1)),
                                                             torch.from_numpy(np.concatenate([gt, gt[:, :2]], 1))))   # (its compute_joint_err is the COCO one)
    je, me = MPII3D.compute_both_err(st, None, None, torch.from_numpy(pred), torch.from_numpy(gt))
    out.update(mpii3d_both_joint=float(je), mpii3d_both_mesh=float(me))
    outs = [dict(joint_coord=pred[n], joint_coord_target=gt[n]) for n in range(N)]
    c = capture("evaluate", ("mpjpe", "pa_mpjpe", "acc_error_h36m"), lambda: MPII3D.evaluate(st, outs))
    out.update(mpii3d_mpjpe=c["mpjpe"].mean(1), mpii3d_pampjpe=c["pa_mpjpe"].mean(1), mpii3d_acc_sum=float(c["acc_error_h36m"]),
               mpii3d_printed=c["printed"])
    print(c["printed"])
    np.savez_compressed(osp.join(HERE, "metrics_pose.npz"), N=N, **out)
    print({k: v for k, v in out.items() if isinstance(v, float)})


if __name__ == "__main__":
    main()
