"""On-device evaluation behind the hot path — drop-ins for the reference's metric code:

* :meth:`Evaluator.compute_both_err`  ==  ``dataset.compute_both_err`` (data/PW3D/dataset.py:269-282), the per-batch running
  MPJPE / MPVPE printed by ``Tester.test`` (lib/core/base.py:227-233), without the D2H copy of ``[B,6890,3]`` meshes;
* :meth:`Evaluator.evaluate`  ==  the arithmetic of ``dataset.evaluate`` (data/PW3D/dataset.py:351-462): MPJPE, PA-MPJPE,
  MPVPE and acceleration error over a (possibly rank-sharded) set of clips, reduced with one small collective; with
  ``gt_joints_mm`` / ``keep_global`` it is ``Human36M.evaluate`` (data/Human36M/dataset.py:715-848): annotated ground-truth
  joints, camera-4 samples only.

Kernels: csrc/metrics.hip through the C ABI.  torch is used for allocation and the final few-float reductions only.  Those
reductions (fp64 sums, cat / stack) may run while pipeline lanes are computing the next batches: torch kernels of exactly these
kinds are ASSERTED bit-correct beside forwards of both product modes on every run of the GPU suite
(tests/test_gpu_bystander.py, part 2).  Clips whose predictions are not finite (non-finite inputs; see
``PMCE.overflowed``) are counted and named in the result (``nonfinite_samples``) instead of silently poisoning the means.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib, assets, sharding

H36M_EVAL_JOINT = (1, 2, 3, 4, 5, 6, 8, 10, 11, 12, 13, 14, 15, 16)   # data/PW3D/dataset.py:35
P = _lib.ptr

# The reference's evaluation flavours: (joint count, evaluated joints, root joint, mesh error reported?)
#   mesh_*   : Tester.test -> dataset.compute_both_err + dataset.evaluate (lib/core/base.py:206-262) - meshes + the 14 H36M eval joints
#   pose_*   : LiftTester.test -> dataset.compute_joint_err + dataset.evaluate_joint (lib/core/base.py:342-387) - lifter output only
#   mpii3d   : config/test_mesh_mpii3d.yml -> MPII3D.compute_both_err / .evaluate (data/MPII3D/dataset.py:539-624): joints only, all 17
FLAVOURS = {
    "mesh_pw3d": dict(joints=17, eval_joint=H36M_EVAL_JOINT, root_joint=0, mesh=True),
    "mesh_h36m": dict(joints=17, eval_joint=H36M_EVAL_JOINT, root_joint=0, mesh=True),       # + camera-4 filter, annotated joints
    "pose_h36m": dict(joints=17, eval_joint=H36M_EVAL_JOINT, root_joint=0, mesh=False),      # Human36M/dataset.py:600-713 (camera 4 only)
    "pose_pw3d": dict(joints=19, eval_joint=tuple(range(19)), root_joint=17, mesh=False),    # PW3D/dataset.py:260-349: COCO set, root [-2] = Pelvis
    "mpii3d": dict(joints=17, eval_joint=tuple(range(17)), root_joint=0, mesh=False),        # MPII3D/dataset.py:539-624
}


class Evaluator:
    def __init__(self, device, j_regressor_h36m=None, root_regressor_row=None, eval_joint=H36M_EVAL_JOINT, root_joint=0):
        """j_regressor_h36m: dense [17,6890] (default: the bundled J_regressor_h36m_correct).  root_regressor_row: dense
        [6890] weights of the joint the MESH is root-aligned with in ``evaluate`` (the SMPL regressor's root row,
        dataset.py:379-384; default = h36m joint 0 when the SMPL model files are unavailable)."""
        self.device = torch.device(device)
        self.lib = _lib.load()
        jr = assets.load_j_regressor("h36m") if j_regressor_h36m is None else np.asarray(j_regressor_h36m)
        self.jr = jr.astype(np.float32)
        # (a root joint beyond the regressor's rows is a joint of ANOTHER joint set - the COCO Pelvis of the pose-only 3DPW flavour - and has no
        # regressor row: that flavour never touches a mesh)
        root = (self.jr[root_joint if root_joint < self.jr.shape[0] else 0] if root_regressor_row is None
                else np.asarray(root_regressor_row, dtype=np.float32))
        self.root_row = root.reshape(1, -1)
        self.rowsum = torch.from_numpy(self.jr.astype(np.float64).sum(1).astype(np.float32)).to(self.device)
        self.eval_idx = torch.tensor(list(eval_joint), dtype=torch.int32, device=self.device)
        self.n_eval, self.root_joint = len(eval_joint), root_joint
        self._max_joint = max(max(eval_joint), root_joint)
        self.flavour, self.mesh_metric, self.n_joints = None, True, None
        # CSR forms of both regressors, built and uploaded ONCE: per_sample runs per batch and must not touch the host
        self._csr_jr = self._upload_csr(self.jr)
        self._csr_root = self._upload_csr(self.root_row)

    @classmethod
    def for_flavour(cls, name: str, device, **kw) -> "Evaluator":
        """The evaluator of one of the reference's test configurations (see FLAVOURS)."""
        f = FLAVOURS[name]
        ev = cls(device, eval_joint=f["eval_joint"], root_joint=f["root_joint"], **kw)
        ev.flavour, ev.mesh_metric, ev.n_joints = name, f["mesh"], f["joints"]
        return ev

    def _upload_csr(self, dense):
        indptr, indices, data = assets.regressor_to_csr(dense)
        return tuple(torch.from_numpy(a).to(self.device) for a in (indptr, indices, data)) + (int(dense.shape[0]),)

    def _regress(self, mesh, csr, scale=1000.0):
        """J_regressor @ (mesh * scale) with a cached CSR (no host work, stream-asynchronous)."""
        ip, ix, dt, rows = csr
        out = torch.empty(mesh.shape[0], rows, 3, device=mesh.device, dtype=torch.float32)
        _lib.check(self.lib.pmce_j_regress_f32(P(mesh), P(ip), P(ix), P(dt), P(out), mesh.shape[0], rows, mesh.shape[1], scale,
                                               _lib.current_stream()), "j_regress")
        return out

    def _sample_errors(self, pm, gm, scale, rp, rg, pj, gj, rowsum, want_joints):
        B, V, _ = pm.shape
        dev = pm.device
        out = [torch.empty(B, device=dev, dtype=torch.float32) for _ in range(3)]
        pe = ge = None
        if want_joints:
            pe = torch.empty(B, self.n_eval, 3, device=dev, dtype=torch.float32)
            ge = torch.empty_like(pe)
        _lib.check(self.lib.pmce_sample_errors_f32(P(pm), P(gm), scale, V, P(rp), P(rg), P(pj), P(gj), pj.shape[1], P(rowsum),
                                                   P(self.eval_idx), self.n_eval, self.root_joint, P(out[0]), P(out[1]),
                                                   P(out[2]), P(pe), P(ge), B, _lib.current_stream()), "sample_errors")
        return out[0], out[1], out[2], pe, ge

    @torch.no_grad()
    def compute_both_err(self, pred_mesh, target_mesh, pred_joint, target_joint):
        """Same arguments and return as the reference method (all in mm, GPU tensors): (joint_mean_error, mesh_mean_error).
        The MPII3D flavour (data/MPII3D/dataset.py:549-558) looks at the joints only and reports a mesh error of 0."""
        if not self.mesh_metric:
            return self.compute_joint_err(pred_joint, target_joint), 0.0
        c = lambda t: t.to(self.device, torch.float32).contiguous()
        mv, mj, _, _, _ = self._sample_errors(c(pred_mesh), c(target_mesh), 1.0, None, None, c(pred_joint), c(target_joint),
                                              None, False)
        return float(mj.mean().item()), float(mv.mean().item())

    # ---- pose-only flavours: the lifter's evaluation (LiftTester.test, lib/core/base.py:342-387) and MPII3D's ----
    @torch.no_grad()
    def joint_errors(self, pred_joint_mm, gt_joint_mm, want_joints=True):
        """Per-sample (mpjpe[N], pampjpe[N], pred_eval_joints[N,n_eval,3], gt_eval_joints[N,n_eval,3]) of joint sets [N,J,3] in mm: root
        alignment by this evaluator's root joint, its evaluated joints, rigid_align in fp64 - sample_errors_kernel with no mesh (V = 0)."""
        c = lambda t: t.to(self.device, torch.float32).contiguous()
        pj, gj = c(pred_joint_mm), c(gt_joint_mm)
        if pj.shape != gj.shape or pj.dim() != 3 or pj.shape[2] != 3:
            raise ValueError(f"joint_errors: pred {tuple(pj.shape)} / target {tuple(gj.shape)} must both be [N, J, 3]")
        if self._max_joint >= pj.shape[1]:
            raise ValueError(f"joint_errors: {pj.shape[1]} joints given, the evaluator's joints reach index {self._max_joint}")
        B, dev = pj.shape[0], pj.device
        mv, mj, pa = (torch.empty(B, device=dev, dtype=torch.float32) for _ in range(3))
        pe = ge = None
        if want_joints:
            pe = torch.empty(B, self.n_eval, 3, device=dev, dtype=torch.float32)
            ge = torch.empty_like(pe)
        _lib.check(self.lib.pmce_sample_errors_f32(None, None, 1.0, 0, None, None, P(pj), P(gj), pj.shape[1], None, P(self.eval_idx),
                                                   self.n_eval, self.root_joint, P(mv), P(mj), P(pa), P(pe), P(ge), B,
                                                   _lib.current_stream()), "sample_errors (joints only)")
        return mj, pa, pe, ge

    @torch.no_grad()
    def compute_joint_err(self, pred_joint, target_joint):
        """== dataset.compute_joint_err (Human36M/dataset.py:600-609, PW3D/dataset.py:260-267, MPII3D/dataset.py:539-547): the running MPJPE
        LiftTester.test prints per batch; joints in mm, GPU tensors [B,J,3]."""
        mj, _, _, _ = self.joint_errors(pred_joint, target_joint, want_joints=False)
        return float(mj.mean().item())

    @torch.no_grad()
    def evaluate_joint(self, pred_joint_mm, gt_joint_mm, seq_ids_global, lo=None, hi=None, keep_global=None):
        """== dataset.evaluate_joint (Human36M/dataset.py:625-713 with keep_global = the camera-4 samples; PW3D/dataset.py:284-349) and
        MPII3D.evaluate (MPII3D/dataset.py:560-624) over a clip set sharded contiguously over the ranks: MPJPE, PA-MPJPE and the
        per-sequence acceleration error of the lifter's (or regressed) joints [hi-lo, J, 3] in mm against the targets."""
        r = RunningEval(self)
        r.add_joints(pred_joint_mm, gt_joint_mm)
        return r.finish(seq_ids_global, lo, hi, keep_global)

    @torch.no_grad()
    def per_sample(self, pred_mesh_m, gt_mesh_m, gt_joints_mm=None):
        """dataset.evaluate's per-sample arithmetic for meshes in METRES (x1000 inside, base.py:223):
        returns (mpvpe[N], mpjpe[N], pampjpe[N], pred_eval_joints[N,14,3], gt_eval_joints[N,14,3]) in mm.
        gt_joints_mm [N,17,3]: annotated ground-truth joints (Human36M/dataset.py:797-799) used instead of the ones
        regressed from the ground-truth mesh."""
        c = lambda t: t.to(self.device, torch.float32).contiguous()
        pm, gm = c(pred_mesh_m), c(gt_mesh_m)
        pj = self._regress(pm, self._csr_jr)
        rp = self._regress(pm, self._csr_root).reshape(-1, 3)
        rg = self._regress(gm, self._csr_root).reshape(-1, 3)
        if gt_joints_mm is None:
            gj = self._regress(gm, self._csr_jr)
        else:
            # The kernel aligns joint k as  j[k] - rowsum[k]*root  (joints regressed from a root-aligned mesh).  Annotated
            # joints are plain coordinates aligned by their own joint 0, so the root term is added here and cancels there.
            gj = c(gt_joints_mm) + self.rowsum[None, :, None] * rg[:, None, :]
        return self._sample_errors(pm, gm, 1000.0, rp, rg, pj, gj.contiguous(), self.rowsum, True)

    @torch.no_grad()
    def accel(self, pe, ge, seq_ids):
        seq = torch.as_tensor(np.asarray(seq_ids), dtype=torch.int32, device=pe.device).contiguous()
        out = torch.empty(pe.shape[0], device=pe.device, dtype=torch.float32)
        _lib.check(self.lib.pmce_accel_error_f32(P(pe.contiguous()), P(ge.contiguous()), P(seq), P(out), pe.shape[0],
                                                 pe.shape[1], _lib.current_stream()), "accel_error")
        return out

    @torch.no_grad()
    def evaluate(self, pred_mesh_m, gt_mesh_m, seq_ids_global, lo=None, hi=None, gt_joints_mm=None, keep_global=None):
        """Metrics over a clip set sharded contiguously over the ranks of the default process group (or unsharded).
        pred/gt: this rank's clips [lo,hi); seq_ids_global: int sequence id of EVERY clip (host array, clip order).
        One all_reduce of 4 floats + one all_gather of the 14x3 eval joints — never the meshes.
        Human3.6M flavour: gt_joints_mm = this rank's annotated joints [hi-lo,17,3]; keep_global = bool per clip (camera 4):
        dropped clips count nowhere, and the acceleration error is taken over the kept clips of each sequence."""
        seq_ids_global = np.asarray(seq_ids_global)
        N = len(seq_ids_global)
        lo = 0 if lo is None else lo
        hi = N if hi is None else hi
        keep_global = np.ones(N, dtype=bool) if keep_global is None else np.asarray(keep_global, dtype=bool)
        mv, mj, pa, pe, ge = self.per_sample(pred_mesh_m, gt_mesh_m, gt_joints_mm)
        assert mv.shape[0] == hi - lo
        k = torch.from_numpy(keep_global[lo:hi]).to(mv.device)
        kd = k.double()
        partial = torch.stack([(mv.double() * kd).sum(), (mj.double() * kd).sum(), (pa.double() * kd).sum(), kd.sum()])
        tot = sharding.reduce_metric_sums(partial)
        allj = sharding.gather_rows(torch.cat([pe, ge], 1))            # [N, 28, 3] in clip order
        kg = torch.from_numpy(keep_global).to(allj.device)
        allj, seq_ids_global = allj[kg], seq_ids_global[keep_global]
        acc = self.accel(allj[:, :self.n_eval].contiguous(), allj[:, self.n_eval:].contiguous(), seq_ids_global)
        n = float(tot[3].item())
        return {"MPVPE": float(tot[0].item()) / n, "MPJPE": float(tot[1].item()) / n, "PA-MPJPE": float(tot[2].item()) / n,
                "ACCEL": float(acc.double().sum().item()) / n, "samples": int(n), **_nonfinite_report(mv, mj, pa, lo)}


def _nonfinite_report(mv, mj, pa, lo):
    """Which of this rank's clips carry non-finite errors (their predictions were inf / nan): count + the first global indices."""
    bad = ~(torch.isfinite(mv) & torch.isfinite(mj) & torch.isfinite(pa))
    nb = int(bad.sum().item())
    idx = (bad.nonzero().flatten()[:16] + lo).tolist() if nb else []
    return {"nonfinite_samples": nb, "nonfinite_first_indices": idx}


class RunningEval:
    """Batch-by-batch form of :meth:`Evaluator.evaluate` for test sets that should not sit in memory as meshes: every batch
    is reduced to its per-sample errors and 14x3 evaluation joints on the device as soon as it is produced (82 KB of mesh
    per clip shrink to 180 B), and the ranks meet once in :meth:`finish`."""

    def __init__(self, evaluator: Evaluator):
        self.ev = evaluator
        self.mv, self.mj, self.pa, self.pe, self.ge = [], [], [], [], []

    @torch.no_grad()
    def add(self, pred_mesh_m, gt_mesh_m, gt_joints_mm=None):
        mv, mj, pa, pe, ge = self.ev.per_sample(pred_mesh_m, gt_mesh_m, gt_joints_mm)
        for lst, t in zip((self.mv, self.mj, self.pa, self.pe, self.ge), (mv, mj, pa, pe, ge)):
            lst.append(t)

    @torch.no_grad()
    def add_joints(self, pred_joint_mm, gt_joint_mm):
        """A batch of the pose-only flavours: joints [B,J,3] in mm, no meshes (MPVPE is reported as None by :meth:`finish`)."""
        mj, pa, pe, ge = self.ev.joint_errors(pred_joint_mm, gt_joint_mm)
        for lst, t in zip((self.mj, self.pa, self.pe, self.ge), (mj, pa, pe, ge)):
            lst.append(t)

    @torch.no_grad()
    def finish(self, seq_ids_global, lo=None, hi=None, keep_global=None):
        """Same result dict as Evaluator.evaluate over this rank's clips [lo, hi) added in order."""
        ev = self.ev
        seq_ids_global = np.asarray(seq_ids_global)
        N = len(seq_ids_global)
        lo = 0 if lo is None else lo
        hi = N if hi is None else hi
        keep_global = np.ones(N, dtype=bool) if keep_global is None else np.asarray(keep_global, dtype=bool)
        if self.mv and len(self.mv) != len(self.mj):
            raise ValueError("RunningEval: batches with meshes (add) and without (add_joints) were mixed")
        mj, pa, pe, ge = (torch.cat(x) for x in (self.mj, self.pa, self.pe, self.ge))
        with_mesh = bool(self.mv)
        mv = torch.cat(self.mv) if with_mesh else torch.zeros_like(mj)
        assert mj.shape[0] == hi - lo, f"added {mj.shape[0]} clips, shard holds {hi - lo}"
        kd = torch.from_numpy(keep_global[lo:hi]).to(mv.device).double()
        tot = sharding.reduce_metric_sums(torch.stack([(mv.double() * kd).sum(), (mj.double() * kd).sum(),
                                                       (pa.double() * kd).sum(), kd.sum()]))
        allj = sharding.gather_rows(torch.cat([pe, ge], 1))
        kg = torch.from_numpy(keep_global).to(allj.device)
        allj, seq = allj[kg], seq_ids_global[keep_global]
        acc = ev.accel(allj[:, :ev.n_eval].contiguous(), allj[:, ev.n_eval:].contiguous(), seq)
        n = float(tot[3].item())
        return {"MPVPE": (float(tot[0].item()) / n) if with_mesh else None, "MPJPE": float(tot[1].item()) / n, "PA-MPJPE": float(tot[2].item()) / n,
                "ACCEL": float(acc.double().sum().item()) / n, "samples": int(n), **_nonfinite_report(mv, mj, pa, lo)}
