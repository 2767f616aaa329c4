#!/usr/bin/env python3
"""Golden vectors for the metric code from the REAL reference functions (build container only):
coord_utils.rigid_align / compute_error_accel (lib/coord_utils.py) and PW3D.compute_both_err / PW3D.evaluate
(data/PW3D/dataset.py), the latter two called as unbound methods on a stub ``self`` (the dataset class itself cannot be
constructed: no annotation files, no SMPL model).  ``evaluate`` only prints 2-decimal summaries, so its per-sample arrays
are captured from its frame locals with a profile hook.  Outputs only; inputs are regenerated from pmce_amd.synth."""
import os, os.path as osp, sys, types, io, contextlib
import numpy as np
import torch

HERE = osp.dirname(osp.abspath(__file__)); REPO = osp.dirname(osp.dirname(HERE)); REF = "/root/reference"
sys.path.insert(0, REPO)
from pmce_amd import synth, assets  # noqa: E402


def shims():
    class AD(dict):
        __getattr__ = dict.__getitem__
    core = types.ModuleType("core"); cc = types.ModuleType("core.config")
    cc.cfg = AD(data_dir="data", TEST=AD(vis=False), vis_dir="/tmp")
    core.config = cc
    pc = types.ModuleType("pycocotools"); pcc = types.ModuleType("pycocotools.coco"); pcc.COCO = object; pc.coco = pcc
    fu = types.ModuleType("funcs_utils"); fu.save_obj = lambda *a, **k: None
    sm = types.ModuleType("smpl"); sm.SMPL = object
    iu = types.ModuleType("_img_utils"); iu.split_into_chunks_mesh = None
    sys.modules.update({"core": core, "core.config": cc, "pycocotools": pc, "pycocotools.coco": pcc, "funcs_utils": fu,
                        "smpl": sm, "_img_utils": iu})
    sys.path.insert(0, osp.join(REF, "lib")); sys.path.insert(0, osp.join(REF, "data"))


def inputs(N=10, seed=3):
    """pred/gt meshes in mm: a smooth base + noise, so errors are O(50 mm) like real evaluations."""
    u = synth.uniform_pm1
    base = u("metrics.base", N * 6890 * 3, seed).reshape(N, 6890, 3) * np.array([350.0, 900.0, 150.0], dtype=np.float32)
    gt = base + u("metrics.gtoff", N * 3, seed).reshape(N, 1, 3) * 200.0
    pred = base + u("metrics.noise", N * 6890 * 3, seed).reshape(N, 6890, 3) * 60.0 + u("metrics.poff", N * 3, seed).reshape(N, 1, 3) * 300.0
    seq = np.array([0, 0, 0, 0, 1, 1, 1, 2, 2, 2][:N])
    return pred.astype(np.float32), gt.astype(np.float32), seq


def smpl_like_regressor(seed=5):
    """stand-in for the (absent) SMPL [24,6890] J_regressor: 24 rows of 8 positive weights each, rows sum to 1."""
    r = np.zeros((24, 6890))
    u = synth.uniform_pm1("metrics.reg", 24 * 8 * 2, seed).reshape(24, 8, 2)
    for j in range(24):
        cols = ((u[j, :, 0] + 1) * 0.5 * 6889).astype(int)
        w = (u[j, :, 1] + 1.5)
        r[j, cols] += w / w.sum()
    return r


def main():
    shims()
    import coord_utils
    from PW3D.dataset import PW3D
    pred, gt, seq = inputs()
    jr = assets.load_j_regressor("h36m").astype(np.float32)
    # --- compute_both_err (called exactly as Tester.test does, base.py:223-227) on torch tensors
    pm, gm = torch.from_numpy(pred), torch.from_numpy(gt)
    J = torch.Tensor(jr)
    pj, gj = torch.matmul(J[None], pm), torch.matmul(J[None], gm)
    stub = types.SimpleNamespace(human36_eval_joint=(1, 2, 3, 4, 5, 6, 8, 10, 11, 12, 13, 14, 15, 16))
    j_err, s_err = PW3D.compute_both_err(stub, pm, gm, pj, gj)
    # --- building blocks
    A = pj[0, list(stub.human36_eval_joint)].numpy().astype(np.float64); B = gj[0, list(stub.human36_eval_joint)].numpy().astype(np.float64)
    aligned = coord_utils.rigid_align(A, B)
    c, R, t = coord_utils.rigid_transform_3D(A, B)
    Aref = A.copy(); Aref[:, 0] *= -1                      # a reflected copy forces the det(R) < 0 branch
    aligned_refl = coord_utils.rigid_align(Aref, B)
    acc = coord_utils.compute_error_accel(joints_gt=gj[:4].numpy().astype(np.float64), joints_pred=pj[:4].numpy().astype(np.float64))
    # --- evaluate(): stub self + frame-local capture
    reg_smpl = smpl_like_regressor()
    N = pred.shape[0]
    st = types.SimpleNamespace(
        vid_indices=[(n, n) for n in range(N)], img_paths=[f"/x/seq{seq[n]}/{n:05d}.jpg" for n in range(N)],
        vid_names=[f"seq{seq[n]}" for n in range(N)], seqlen=16, smpl_vertex_num=6890, smpl_root_joint_idx=0,
        joint_regressor_smpl=reg_smpl, human36_root_joint_idx=0, human36_eval_joint=stub.human36_eval_joint,
        coco_joints_name=('Nose', 'L_Eye', 'R_Eye', 'L_Ear', 'R_Ear', 'L_Shoulder', 'R_Shoulder', 'L_Elbow', 'R_Elbow', 'L_Wrist',
                          'R_Wrist', 'L_Hip', 'R_Hip', 'L_Knee', 'R_Knee', 'L_Ankle', 'R_Ankle', 'Pelvis', 'Neck'),
        mesh_model=types.SimpleNamespace(joint_regressor_h36m=assets.load_j_regressor("h36m").astype(np.float32),
                                         joint_regressor_coco=assets.load_j_regressor("coco").astype(np.float32), face=None))
    outs = [dict(mesh_coord=pred[n], mesh_coord_target=gt[n]) for n in range(N)]
    captured = {}

    def prof(frame, event, arg):
        if event == "return" and frame.f_code.co_name == "evaluate":
            for k in ("mpjpe_h36m", "pampjpe_h36m", "mpvpe", "acc_error_h36m"):
                captured[k] = np.array(frame.f_locals[k], dtype=np.float64)
    sys.setprofile(prof)
    with contextlib.redirect_stdout(io.StringIO()) as buf:
        PW3D.evaluate(st, outs)
    sys.setprofile(None)
    np.savez_compressed(osp.join(HERE, "metrics.npz"), N=N, seed=3, seq=seq, j_err=float(j_err), s_err=float(s_err),
                        aligned=aligned, c=c, R=R, t=t, aligned_refl=aligned_refl, accel4=acc,
                        mpjpe=captured["mpjpe_h36m"], pampjpe=captured["pampjpe_h36m"],
                        mpvpe_mean_per_sample=captured["mpvpe"].mean(1), mpvpe_first=captured["mpvpe"][0, :64],
                        acc_error_sum=float(captured["acc_error_h36m"]), printed=buf.getvalue())
    print("compute_both_err:", float(j_err), float(s_err)); print(buf.getvalue())


if __name__ == "__main__":
    main()
