"""Two-stream wrapper façade — drop-in for reference lib/models/PMCE.py."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _lib, assets, packing, synth
from ..config import FEAT_DIM, NUM_VERTS_FULL, SEQLEN, cfg
from ..runtime import HipEngine, HipModuleBase, _check_input, build_param_tree


class PMCE(HipModuleBase):
    """forward(pose2d[B,16,J,2], img_feat[B,16,2048]) -> (cam_mesh[B,6890,3] m, cam_pose[B,J,3] m, pose3d[B,J,3] mm)
    — same order as the reference (PMCE.py:15-20).  State-dict keys: ``pose_lifter.*``, ``pose_mesh_coevo.*``."""

    def __init__(self, num_joint, embed_dim, depth, base_data_dir=None):
        super().__init__()
        self.num_joint, self.embed_dim, self.depth = num_joint, embed_dim, depth
        build_param_tree(self, synth.pmce_spec(num_joint, embed_dim, depth))
        v431, vj, src = assets.build_template(base_data_dir)
        self.pose_mesh_coevo.init_vertices.copy_(torch.from_numpy(v431))
        self.vj_relation = vj
        self.base_data_source = src
        self._j_regressor = None
        self.eval()

    def set_j_regressor(self, j_regressor):
        """Register the caller's regressor (``Tester.J_regressor``, lib/core/base.py:196) so that
        :meth:`forward_with_joints` also returns ``J_regressor @ (cam_mesh*1000)`` (base.py:223-225)."""
        self._j_regressor = np.asarray(j_regressor)
        self._dirty = True

    def _build_engine(self, dev):
        eng = HipEngine(self.num_joint, self.embed_dim, self.depth)
        sd = self.state_dict()
        eng.register(packing.pack_lifter(sd, "pose_lifter.", dev, self.num_joint, self.embed_dim, self.depth))
        eng.register(packing.pack_decoder(sd, "pose_mesh_coevo.", dev, self.num_joint, self.vj_relation))
        if self._j_regressor is not None:
            eng.set_regressor(self._j_regressor)
        eng.finalize()
        return eng

    def _run(self, pose2d, img_feat, want_joints, eng=None):
        eng = eng or self._ensure_packed()
        pose2d = _check_input(pose2d, (SEQLEN, self.num_joint, 2), "pose2d")
        img_feat = _check_input(img_feat, (SEQLEN, FEAT_DIM), "img_feat")
        B = pose2d.shape[0]
        dev = pose2d.device
        mesh = torch.empty(B, NUM_VERTS_FULL, 3, device=dev, dtype=torch.float32)
        pose = torch.empty(B, self.num_joint, 3, device=dev, dtype=torch.float32)
        pose3d = torch.empty(B, self.num_joint, 3, device=dev, dtype=torch.float32)
        pred = None
        if want_joints:
            if not eng.regressor_rows:
                raise _lib.PmceError("forward_with_joints needs set_j_regressor(...) first")
            pred = torch.empty(B, eng.regressor_rows, 3, device=dev, dtype=torch.float32)
        if B == 0:               # an empty batch is an empty result (the reference's modules return empty tensors too)
            return mesh, pose, pose3d, pred
        ws = eng.workspace(B)
        _lib.check(eng.lib.pmce_forward(eng.handle, _lib.ptr(pose2d), _lib.ptr(img_feat), _lib.ptr(mesh), _lib.ptr(pose),
                                        _lib.ptr(pose3d), _lib.ptr(pred), B, C.c_void_p(ws.data_ptr()), ws.numel(),
                                        _lib.current_stream()), "pmce_forward")
        return mesh, pose, pose3d, pred

    @torch.no_grad()
    def forward(self, pose2d, img_feat):
        """The reference's call (PMCE.py:15-20; ``model(pose2d, img_feat)`` at lib/core/base.py:222), under the module's overflow policy
        (default "rerun": see HipModuleBase.set_overflow_policy)."""
        mesh, pose, pose3d, _ = self._guarded(lambda eng: self._run(pose2d, img_feat, False, eng))
        return mesh, pose, pose3d

    @torch.no_grad()
    def forward_with_joints(self, pose2d, img_feat):
        """(cam_mesh, cam_pose, pose3d, pred_pose_mm) - the forward plus the caller's tail of Tester.test; same overflow policy."""
        return self._guarded(lambda eng: self._run(pose2d, img_feat, True, eng))

    @torch.no_grad()
    def forward_checked(self, pose2d, img_feat, want_joints: bool = False):
        """``forward`` / ``forward_with_joints`` with the "rerun" behaviour whatever the module's policy, saying whether it happened:
        returns (outputs, reran)."""
        eng = self._ensure_packed()
        want = want_joints and eng.regressor_rows > 0
        out = self._run(pose2d, img_feat, want)
        torch.cuda.synchronize(eng.device)
        if not eng.overflowed():
            return (out if want_joints else out[:3]), False
        eng.clear_overflow()
        out = eng.run_on_f32_pipe(lambda e: self._run(pose2d, img_feat, want, e))
        torch.cuda.synchronize(eng.device)
        return (out if want_joints else out[:3]), True

    # benchmarking hooks
    def set_concurrency(self, enable=True):
        """Two-stream execution of independent branches inside one forward (default on)."""
        self._ensure_packed().set_concurrency(enable)

    def profile(self, enable=True):
        """Per-kernel-class HIP-event timing; kernels are serialised on one stream while it is on so that each
        launch is priced alone."""
        eng = self._ensure_packed()
        eng.set_concurrency(not enable)
        eng.profile(enable)

    def profile_read(self):
        return self._ensure_packed().profile_read()

    def pipeline(self, depth: int = 2, stagger=None, on_overflow: str = "warn") -> "Pipeline":
        """Several batches in flight at once on shared weights; see :class:`Pipeline`."""
        return Pipeline(self, depth, stagger, on_overflow)

    def graphed(self, batch: int, want_joints: bool = True) -> "GraphedForward":
        """The forward for a fixed small batch captured once as a hipGraph; see :class:`GraphedForward`."""
        return GraphedForward(self, batch, want_joints)


class GraphedForward:
    """``forward_with_joints`` for ONE batch size, captured as a hipGraph and replayed: the ~110 launches of a forward (and
    its internal fork/join onto the side stream) become one graph launch.  For the launch-bound small batches of the
    reference's demo (batch 1, main/run_demo.py:332,145); results are bit-identical to the eager call.

        gf = model.graphed(1)
        mesh, pose, pose3d, pred = gf(pose2d, img_feat)     # outputs are the graph's static buffers: consume or clone
                                                            # them before the next call
    """

    def __init__(self, model: "PMCE", batch: int, want_joints: bool = True):
        main = model._ensure_packed()
        dev = main.device
        self.model, self._main, self.batch = model, main, batch
        self.eng = main.clone_shared()           # a handle of its own: the graph owns its workspace, side stream and events
        want = want_joints and self.eng.regressor_rows > 0
        self.pose2d = torch.zeros(batch, SEQLEN, model.num_joint, 2, device=dev)
        self.img_feat = torch.zeros(batch, SEQLEN, FEAT_DIM, device=dev)
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s), torch.no_grad():
            for _ in range(2):                   # outside capture: workspace allocation, side stream and event creation
                model._run(self.pose2d, self.img_feat, want, self.eng)
        s.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=s), torch.no_grad():
            self.outputs = model._run(self.pose2d, self.img_feat, want, self.eng)

    @torch.no_grad()
    def __call__(self, pose2d, img_feat):
        if self.model._engine is not self._main or self.model._dirty:
            raise _lib.PmceError("the model's weights were re-packed after this graph was captured: call model.graphed() again")
        if tuple(pose2d.shape) != tuple(self.pose2d.shape) or tuple(img_feat.shape) != tuple(self.img_feat.shape):
            raise ValueError(f"this graph was captured for batch {self.batch}: got {tuple(pose2d.shape)} / {tuple(img_feat.shape)}")
        self.pose2d.copy_(pose2d)
        self.img_feat.copy_(img_feat)
        self.graph.replay()
        return self.outputs


class Pipeline:
    """Keeps ``depth`` forwards of independent batches in flight (clips are independent, reference lib/_img_utils.py:74-78):
    batch k runs on lane k % depth - a handle of its own on the shared weights, with its own workspace and streams - so the
    decoder's latency-bound kernels of one batch overlap the matrix-bound pose lifter of the next.  Results are identical to
    ``model.forward_with_joints`` (same kernels, same order within a batch).

        pipe = model.pipeline(depth=2)
        tickets = [pipe.submit(p, f) for p, f in batches]      # returns at once
        for t in tickets:
            mesh, pose, pose3d, pred = t.result()             # makes the CURRENT stream wait for that batch
    """

    class Ticket:
        def __init__(self, outputs, done, index=-1, inputs=None, lane=0):
            self.outputs, self.done, self.index, self.inputs, self.lane = outputs, done, index, inputs, lane

        def result(self):
            cur = torch.cuda.current_stream(self.outputs[0].device)
            cur.wait_event(self.done)
            for t in self.outputs:              # allocated on the lane's stream, consumed on the caller's
                if t is not None:
                    t.record_stream(cur)
            return self.outputs

    def __init__(self, model: "PMCE", depth: int = 2, stagger=None, on_overflow: str = "warn"):
        """on_overflow: what :meth:`synchronize` does when a product reported a non-finite value - "warn" (default: name the batches),
        "raise", or "rerun" (compute them again on the fp32 pipe into the same output tensors).  Only "rerun" keeps references to a
        batch's INPUT tensors (the last 4 x depth submits) - the caller must then leave them unmodified until the next synchronize();
        otherwise the pipeline holds the recent tickets weakly and retains nothing the caller has dropped.
        stagger: start a batch's pose lifter only when the previous batch's has finished (pmce_model_wait_lifter), so that it is lifter(k+1)
        that runs beside decoder(k).  None (default) decides per submit from the batch size (:meth:`staggers`): free-running lanes are faster for
        small and medium batches (+9 % at B = 64, +5 % at J = 19 / B = 128), staggered ones when a large batch shares the GPU with other streams
        (B = 256: equal back to back, +8 % through the evaluation harness with its metric kernels, +2.6 % host-fed:
        profiles/r06_v_lanes_staggered_vs_free.txt).  True / False force one form."""
        self.stagger, self.prev = stagger, None
        if depth < 1:
            raise ValueError("depth must be >= 1")
        if on_overflow not in ("warn", "raise", "rerun"):
            raise ValueError("on_overflow must be 'warn', 'raise' or 'rerun'")
        self.model, self.depth, self.on_overflow = model, depth, on_overflow
        self.engines, self.streams, self._main = [], [], None
        self.k = 0
        import collections
        self._recent = collections.deque(maxlen=4 * depth)   # tickets a drain can still check (weak references unless on_overflow == "rerun")
        self.reran = []                                       # indices of batches a drain re-ran on the fp32 pipe
        self._bind()

    STAGGER_FROM_BATCH = 192

    def staggers(self, batch: int) -> bool:
        """Whether a submit of ``batch`` clips waits for the previous batch's lifter."""
        return bool(self.stagger) if self.stagger is not None else batch >= Pipeline.STAGGER_FROM_BATCH

    def _bind(self):
        """(Re)build the lanes on the model's current packed weights (they are re-packed after load_state_dict / .to())."""
        main = self.model._ensure_packed()
        if not self.engines or self._main is not main:
            # every lane is a handle of its own (the model's handle and workspace stay free for direct forward() calls)
            self._main = main
            self.engines = [main.clone_shared() for _ in range(self.depth)]
            if main.gemm_mode() == "split_f16" and not _lib.split_overlap():
                # diagnostic (PMCE_SPLIT_OVERLAP=0): the lanes keep their own handles and workspaces (batches are enqueued ahead
                # of the GPU) but run down ONE stream, in order
                st = torch.cuda.Stream(device=main.device)
                self.streams = [st] * self.depth
            else:
                self.streams = [torch.cuda.Stream(device=main.device) for _ in range(self.depth)]
            self.prev = None

    @torch.no_grad()
    def submit(self, pose2d, img_feat, want_joints: bool = True) -> "Pipeline.Ticket":
        self._bind()
        lane = self.k % len(self.engines)
        self.k += 1
        st = self.streams[lane]
        cur = torch.cuda.current_stream(pose2d.device)
        ready = torch.cuda.Event()
        ready.record(cur)                      # inputs produced on the caller's stream
        st.wait_event(ready)
        if self.staggers(pose2d.shape[0]) and self.prev is not None and self.prev is not self.engines[lane]:
            # start this batch's pose lifter when the previous batch's has finished: lifter(k+1) overlaps decoder(k)
            _lib.check(self.engines[lane].lib.pmce_model_wait_lifter(self.prev.handle, C.c_void_p(st.cuda_stream)), "model_wait_lifter")
        self.prev = self.engines[lane]
        with torch.cuda.stream(st):
            out = self.model._run(pose2d, img_feat, want_joints and self.engines[lane].regressor_rows > 0, self.engines[lane])
            pose2d.record_stream(st)
            img_feat.record_stream(st)
            done = torch.cuda.Event()
            done.record(st)
        keep = self.on_overflow == "rerun"
        t = Pipeline.Ticket(out, done, self.k - 1, (pose2d, img_feat, want_joints) if keep else None, lane)
        import weakref
        self._recent.append(t if keep else weakref.ref(t))
        return t

    def prepare(self, batch: int):
        """One-time setup of every lane for ``batch`` clips, so that no lane pays it at its first real submit: workspace
        allocation, creation of the lane's internal stream and events, first touch of its buffers (one forward on zeros)."""
        self._bind()
        J = self.model.num_joint
        dev = self.engines[0].device
        p = torch.zeros(batch, SEQLEN, J, 2, device=dev)
        f = torch.zeros(batch, SEQLEN, FEAT_DIM, device=dev)
        for eng, st in zip(self.engines, self.streams):
            st.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(st):
                self.model._run(p, f, eng.regressor_rows > 0, eng)
                p.record_stream(st)
                f.record_stream(st)
        self.synchronize()
        self.prev = None
        return self

    def synchronize(self, on_overflow: str = None):
        """Wait for every lane, then poll the model's overflow word.  If a product reported a non-finite value, the recent batches
        (of the last 4 x depth submits, those whose tickets are still alive) whose outputs are not finite are named in a warning
        ("warn"), or PmceError is raised ("raise"), or - "rerun", which the pipeline must have been CREATED with, because only then
        does it hold the inputs - they are computed again on the fp32 matrix pipe INTO THE SAME OUTPUT TENSORS.  The word is cleared
        either way (it is a report, not a lock).  Returns the indices of the named batches."""
        on_overflow = on_overflow or self.on_overflow
        if on_overflow == "rerun" and self.on_overflow != "rerun":
            raise ValueError("Pipeline.synchronize(on_overflow='rerun') needs a pipeline created with on_overflow='rerun' (it holds no inputs otherwise)")
        for st in self.streams:
            st.synchronize()
        main = self._main
        if main is None or not (main.overflowed() or self.model._overflow_carried):     # (carried: a "rerun" forward() took an earlier report
            self._recent.clear()                                                         #  out of the word - it is still this drain's to name)
            return []
        self.model._overflow_carried = False
        recent = [t for t in ((r if isinstance(r, Pipeline.Ticket) else r()) for r in self._recent) if t is not None]
        bad = [t for t in recent if not all(bool(torch.isfinite(o).all()) for o in t.outputs if o is not None)]
        main.clear_overflow()
        names = [t.index for t in bad]
        msg = (f"pmce_amd.Pipeline: a product of the split-f16 form produced non-finite values; batches {names} of the last "
               f"{len(recent)} live tickets hold non-finite outputs" + ("" if bad else " (none of those: an earlier or already dropped batch)"))
        if on_overflow == "raise":
            self._recent.clear()
            raise _lib.PmceError(msg)
        import warnings
        if on_overflow == "rerun":
            for t in bad:
                p2, f, want = t.inputs
                eng = self.engines[t.lane]
                new = eng.run_on_f32_pipe(lambda e: self.model._run(p2, f, want and e.regressor_rows > 0, e))
                for dst, src in zip(t.outputs, new):
                    if dst is not None:
                        dst.copy_(src)
            torch.cuda.synchronize(main.device)
            main.clear_overflow()
            self.reran += names
            msg += "; they were computed again on the fp32 matrix pipe (non-finite INPUTS stay non-finite, as in the reference)"
        warnings.warn(msg)
        self._recent.clear()
        return names


def get_model(num_joint, embed_dim, depth):
    """Same signature as reference PMCE.get_model (PMCE.py:23-26)."""
    return PMCE(num_joint, embed_dim, depth)
