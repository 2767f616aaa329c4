"""ORACLE (test infrastructure only) for the evaluation metrics that sit directly after the hot path — SURVEY §8f
rank 1.  numpy float64 restatement of:
  * ``compute_both_err`` (reference data/PW3D/dataset.py:269-282, identical in data/Human36M/dataset.py:611-623),
  * ``rigid_transform_3D`` / ``rigid_align`` (lib/coord_utils.py:151-173),
  * ``compute_error_accel`` (lib/coord_utils.py:218-245),
  * the per-sample arithmetic of ``PW3D.evaluate`` (data/PW3D/dataset.py:351-462) and of ``Human36M.evaluate``
    (data/Human36M/dataset.py:715-848: camera-4 samples only, ANNOTATED ground-truth joints).
Pinned against outputs of the reference's own functions: tests/golden/metrics.npz, metrics_h36m.npz
(tests/golden/make_golden_metrics.py, make_golden_metrics_h36m.py).
"""
import numpy as np

H36M_EVAL_JOINT = (1, 2, 3, 4, 5, 6, 8, 10, 11, 12, 13, 14, 15, 16)   # data/PW3D/dataset.py:35


def compute_both_err(pred_mesh, target_mesh, pred_joint, target_joint, eval_joint=H36M_EVAL_JOINT):
    """dataset.py:269-282: root-align by regressed joint 0, mean per-vertex / per-eval-joint L2 over the batch."""
    pred_mesh, target_mesh = pred_mesh - pred_joint[:, :1, :], target_mesh - target_joint[:, :1, :]
    pred_joint, target_joint = pred_joint - pred_joint[:, :1, :], target_joint - target_joint[:, :1, :]
    pj, tj = pred_joint[:, eval_joint, :], target_joint[:, eval_joint, :]
    mesh_mean_error = np.sqrt(((pred_mesh - target_mesh) ** 2).sum(axis=2)).mean()
    joint_mean_error = np.sqrt(((pj - tj) ** 2).sum(axis=2)).mean()
    return joint_mean_error, mesh_mean_error


def rigid_transform_3D(A, B):
    """coord_utils.py:151-167 (similarity Procrustes: scale c, rotation R, translation t mapping A onto B)."""
    n, dim = A.shape
    centroid_A = np.mean(A, axis=0)
    centroid_B = np.mean(B, axis=0)
    H = np.dot(np.transpose(A - centroid_A), B - centroid_B) / n
    U, s, V = np.linalg.svd(H)
    R = np.dot(np.transpose(V), np.transpose(U))
    if np.linalg.det(R) < 0:
        s[-1] = -s[-1]
        V[2] = -V[2]
        R = np.dot(np.transpose(V), np.transpose(U))
    varP = np.var(A, axis=0).sum()
    c = 1 / varP * np.sum(s)
    t = -np.dot(c * R, np.transpose(centroid_A)) + np.transpose(centroid_B)
    return c, R, t


def rigid_align(A, B):
    """coord_utils.py:170-173."""
    c, R, t = rigid_transform_3D(A, B)
    return np.transpose(np.dot(c * R, np.transpose(A))) + t


def compute_error_accel(joints_gt, joints_pred):
    """coord_utils.py:218-245 with vis=None: (N-2) per-frame mean acceleration error."""
    accel_gt = joints_gt[:-2] - 2 * joints_gt[1:-1] + joints_gt[2:]
    accel_pred = joints_pred[:-2] - 2 * joints_pred[1:-1] + joints_pred[2:]
    normed = np.linalg.norm(accel_pred - accel_gt, axis=2)
    return np.mean(normed, axis=1)


def evaluate_samples(mesh_out, mesh_gt, reg_root, root_idx, reg_h36m, seq_ids, eval_joint=H36M_EVAL_JOINT, gt_joints=None,
                     keep=None):
    """Per-sample arithmetic of PW3D.evaluate (dataset.py:372-433) for meshes already in millimetres.
    reg_root[Rr,6890] with row root_idx = the SMPL regressor's root joint (dataset.py:379-384); reg_h36m[17,6890].
    Human36M.evaluate (Human36M/dataset.py:715-848) is the same arithmetic with two differences: only samples with
    keep[n] (camera 4, :728-730,760-762) take part, and the ground-truth joints are the annotated gt_joints[N,17,3]
    (:797-799) instead of the ones regressed from the ground-truth mesh.
    Returns dict(mpvpe[N,6890], mpjpe[N,14], pampjpe[N,14], accel_sum, summary means)."""
    if keep is not None:
        keep = np.asarray(keep, dtype=bool)
        mesh_out, mesh_gt, seq_ids = mesh_out[keep], mesh_gt[keep], np.asarray(seq_ids)[keep]
        gt_joints = None if gt_joints is None else np.asarray(gt_joints)[keep]
    N = mesh_out.shape[0]
    mpvpe = np.zeros((N, mesh_out.shape[1]))
    mpjpe = np.zeros((N, len(eval_joint)))
    pampjpe = np.zeros((N, len(eval_joint)))
    P, G = [], []
    for n in range(N):
        mo, mg = mesh_out[n].astype(np.float64), mesh_gt[n].astype(np.float64)
        mo = mo - np.dot(reg_root, mo)[root_idx]                      # :379-384 root joint alignment
        mg = mg - np.dot(reg_root, mg)[root_idx]
        mpvpe[n] = np.sqrt(np.sum((mo - mg) ** 2, 1))                 # :389
        po = np.dot(reg_h36m, mo); po = po - po[0]; po = po[eval_joint, :]   # :392-394
        pg = np.dot(reg_h36m, mg) if gt_joints is None else gt_joints[n].astype(np.float64)
        pg = pg - pg[0]; pg = pg[eval_joint, :]                      # :395-397
        mpjpe[n] = np.sqrt(np.sum((po - pg) ** 2, 1))                 # :431
        pampjpe[n] = np.sqrt(np.sum((rigid_align(po, pg) - pg) ** 2, 1))  # :432-433
        P.append(po); G.append(pg)
    P, G = np.array(P), np.array(G)
    acc = 0.0                                                         # :415-429,444-449: per sequence, ends count as 0
    start = 0
    for n in range(1, N + 1):
        if n == N or seq_ids[n] != seq_ids[start]:
            L = n - start
            a = np.zeros(L)
            if L >= 3:
                a[1:-1] = compute_error_accel(joints_pred=P[start:n], joints_gt=G[start:n])
            acc += np.mean(a) * L
            start = n
    return dict(mpvpe=mpvpe, mpjpe=mpjpe, pampjpe=pampjpe, MPJPE=np.mean(mpjpe), PA_MPJPE=np.mean(pampjpe),
                MPVPE=np.mean(mpvpe), ACCEL=acc / N, pred_j=P, gt_j=G)


# ---- pose-only flavours (lifter evaluation, LiftTester.test lib/core/base.py:342-387) and MPII3D ------------------------------------------
# (joint set, root joint, evaluated joints) of the three reference implementations:
POSE_FLAVOURS = {
    "pose_h36m": dict(root=0, eval_joint=H36M_EVAL_JOINT),         # Human36M/dataset.py:600-609,625-713 (+ camera-4 filter in evaluate_joint)
    "pose_pw3d": dict(root=-2, eval_joint=None),                   # PW3D/dataset.py:260-267,284-349: COCO set (19), root [-2:-1], every joint
    "mpii3d": dict(root=0, eval_joint=None),                       # MPII3D/dataset.py:539-547,560-624: 17 joints, root 0, every joint
}


def compute_joint_err(pred_joint, target_joint, root=0, eval_joint=None):
    """dataset.compute_joint_err: root-align both joint sets [B,J,3], (H36M: keep the eval joints,) mean per-joint L2 over the batch."""
    root = root % pred_joint.shape[1]
    pj, tj = pred_joint - pred_joint[:, root:root + 1, :], target_joint - target_joint[:, root:root + 1, :]
    if eval_joint is not None:
        pj, tj = pj[:, eval_joint, :], tj[:, eval_joint, :]
    return np.sqrt(((pj - tj) ** 2).sum(axis=2)).mean()


def evaluate_joint_samples(pred_j, gt_j, seq_ids, root=0, eval_joint=None, keep=None):
    """Per-sample arithmetic of dataset.evaluate_joint / MPII3D.evaluate for joint sets [N,J,3] in mm: root alignment, (eval joints,) MPJPE,
    PA-MPJPE after rigid_align, acceleration error per sequence with the end samples counted as 0.  keep: Human36M.evaluate_joint's camera-4
    samples (:640-642,663-665) - the others take part nowhere."""
    pred_j, gt_j, seq_ids = np.asarray(pred_j, np.float64), np.asarray(gt_j, np.float64), np.asarray(seq_ids)
    if keep is not None:
        keep = np.asarray(keep, dtype=bool)
        pred_j, gt_j, seq_ids = pred_j[keep], gt_j[keep], seq_ids[keep]
    N, J = pred_j.shape[:2]
    root = root % J
    ej = list(range(J)) if eval_joint is None else list(eval_joint)
    mpjpe, pampjpe = np.zeros((N, len(ej))), np.zeros((N, len(ej)))
    P, G = [], []
    for n in range(N):
        po, pg = pred_j[n] - pred_j[n][root:root + 1], gt_j[n] - gt_j[n][root:root + 1]
        po, pg = po[ej, :], pg[ej, :]
        mpjpe[n] = np.sqrt(np.sum((po - pg) ** 2, 1))
        pampjpe[n] = np.sqrt(np.sum((rigid_align(po, pg) - pg) ** 2, 1))
        P.append(po); G.append(pg)
    P, G = np.array(P), np.array(G)
    acc, start = 0.0, 0
    for n in range(1, N + 1):
        if n == N or seq_ids[n] != seq_ids[start]:
            L = n - start
            a = np.zeros(L)
            if L >= 3:
                a[1:-1] = compute_error_accel(joints_pred=P[start:n], joints_gt=G[start:n])
            acc += np.mean(a) * L
            start = n
    return dict(mpjpe=mpjpe, pampjpe=pampjpe, MPJPE=np.mean(mpjpe), PA_MPJPE=np.mean(pampjpe), ACCEL=acc / N, acc_sum=acc, pred_j=P, gt_j=G)
