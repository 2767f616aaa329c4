"""Input staging either side of the hot path: detector output -> model input, window tables for a
multi-video dataset, and the host -> device feed.

  * ``prepare_pose2d``: per-frame keypoints (pixels) -> pose2d table [L, J, 2] on the GPU: pelvis/neck appended and the
    screen normalisation of data/PW3D/dataset.py:185-204 in one kernel (``pmce_prepare_pose2d_f32``).
  * ``mesh_window_table``: ``split_into_chunks_mesh`` (lib/_img_utils.py:58-92) - windows of every video of a frame list.
  * ``PinnedFeeder``: pinned host ring buffers + a copy stream, so that batch k+1 crosses PCIe while batch k computes.  The
    per-frame tables (8 KB of features per frame) are what should cross the bus; windows are assembled on the GPU
    (streaming.assemble_windows), not on the host, where every frame would be copied 16 times.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib

COCO_JOINTS = ('Nose', 'L_Eye', 'R_Eye', 'L_Ear', 'R_Ear', 'L_Shoulder', 'R_Shoulder', 'L_Elbow', 'R_Elbow', 'L_Wrist',
               'R_Wrist', 'L_Hip', 'R_Hip', 'L_Knee', 'R_Knee', 'L_Ankle', 'R_Ankle')


def prepare_pose2d(keypoints: torch.Tensor, img_shapes: torch.Tensor, joints_name=COCO_JOINTS, extra: int = 2) -> torch.Tensor:
    """keypoints [L, J0, >=2] (x, y[, score]) in pixels and img_shapes int[L, 2] = (height, width), both on the GPU ->
    float32 [L, J0 + extra, 2] normalised screen coordinates.  extra = 2 appends pelvis and neck (the COCO-19 input of the
    3DPW checkpoints, dataset.py:160-161), 1 only the pelvis, 0 nothing."""
    lib = _lib.load()
    kp = keypoints.to(torch.float32).contiguous()
    sh = img_shapes.to(torch.int32).contiguous()
    L, J0, D = kp.shape
    if sh.shape != (L, 2):
        raise ValueError(f"img_shapes must be [{L}, 2] (height, width), got {tuple(sh.shape)}")
    idx = {n: joints_name.index(n) for n in ('L_Hip', 'R_Hip', 'L_Shoulder', 'R_Shoulder')} if extra else \
        {'L_Hip': 0, 'R_Hip': 0, 'L_Shoulder': 0, 'R_Shoulder': 0}
    out = torch.empty(L, J0 + extra, 2, device=kp.device, dtype=torch.float32)
    _lib.check(lib.pmce_prepare_pose2d_f32(_lib.ptr(kp), D, _lib.ptr(sh), _lib.ptr(out), L, J0, extra, idx['L_Hip'], idx['R_Hip'],
                                           idx['L_Shoulder'], idx['R_Shoulder'], _lib.current_stream()), "prepare_pose2d")
    return out


def mesh_window_table(img_names, seqlen: int = 16, stride: int = 1, mid_valid=None, match_vibe: bool = True) -> np.ndarray:
    """[start, end] (inclusive indices into ``img_names``) of every window of every video, as ``split_into_chunks_mesh``
    builds them (lib/_img_utils.py:58-92).  A frame's video is its path minus the last 11 characters
    ('image_00012.jpg'); videos are visited in order of first appearance and must be contiguous in the (sorted) list;
    videos shorter than ``seqlen`` yield nothing; windows whose middle frame has ``mid_valid == False`` are dropped; for
    stride != seqlen and match_vibe the trailing windows past the last full 16-frame chunk are dropped."""
    n = len(img_names)
    mid_valid = np.ones(n, dtype=bool) if mid_valid is None else np.asarray(mid_valid, dtype=bool)
    vid = np.array([name[:-11] for name in img_names])
    _, first = np.unique(vid, return_index=True)
    bounds = np.append(np.sort(first), n)
    rows = []
    for a, b in zip(bounds[:-1], bounds[1:]):
        length = b - a
        if length < seqlen:
            continue
        starts = np.arange(a, b - seqlen + 1, stride)
        starts = starts[mid_valid[starts + seqlen // 2]]
        if len(starts) == 0:
            continue
        ends = starts + seqlen - 1
        if stride != seqlen and match_vibe:
            vibe_last_end = a + (length // 16) * 16 - 1
            hit = np.nonzero(ends == vibe_last_end)[0]
            if len(hit):                       # the last window ending on the VIBE boundary closes the list
                starts, ends = starts[: hit[-1] + 1], ends[: hit[-1] + 1]
        rows.append(np.stack([starts, ends], 1))
    return np.concatenate(rows).astype(np.int64) if rows else np.zeros((0, 2), dtype=np.int64)


def pose_window_table(img_names, seqlen: int = 16, stride: int = 1, match_vibe: bool = True) -> np.ndarray:
    """The window list of the pose-only path, as ``split_into_chunks_pose`` builds it (lib/_img_utils.py:27-55): the same
    per-video stride-``stride`` windows and VIBE-tail rule as :func:`mesh_window_table`, without the middle-frame validity
    filter.  For a single video it equals :func:`pmce_amd.streaming.window_indices`."""
    return mesh_window_table(img_names, seqlen, stride, None, match_vibe)


class PinnedFeeder:
    """Double-buffered host -> device feed: ``slots`` pinned host buffers per tensor and a dedicated copy stream.

        feeder = PinnedFeeder(device, {"pose2d": ((B, 16, J, 2), torch.float32), "img_feat": ((B, 16, 2048), torch.float32)})
        for dev_batch in feeder.run(host_batches):      # host_batches yields dicts of numpy arrays / CPU tensors
            model(dev_batch["pose2d"], dev_batch["img_feat"])

    ``run`` keeps ``slots - 1`` copies in flight ahead of the batch it yields; the compute stream waits on the copy's
    event (no host synchronisation), and a slot is re-filled only after the compute stream has passed the batch that
    used it."""

    def __init__(self, device, spec: dict, slots: int = 2):
        self.device = torch.device(device)
        self.slots = slots
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.host = [{k: torch.empty(shape, dtype=dt).pin_memory() for k, (shape, dt) in spec.items()} for _ in range(slots)]
        self.host_np = [{k: t.numpy() for k, t in h.items()} for h in self.host]
        self.dev = [{k: torch.empty(shape, dtype=dt, device=self.device) for k, (shape, dt) in spec.items()} for _ in range(slots)]
        self.copied = [torch.cuda.Event() for _ in range(slots)]
        self.consumed = [torch.cuda.Event() for _ in range(slots)]
        self._used = [False] * slots
        # the first DMA out of a fresh pinned buffer costs ~10 ms (page registration on first use): pay it here, not in the
        # first batches
        # The device buffers come from the CURRENT stream's pool: the allocator may hand out blocks whose previous owner's kernels
        # are still queued on that stream (freed tensors are reusable at once by the same stream).  The copy stream is ordered
        # behind everything queued there so far, or a late kernel of a dead tensor would write into a batch already copied in.
        self.copy_stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.copy_stream):
            for h, d in zip(self.host, self.dev):
                for k in h:
                    d[k].copy_(h[k].zero_(), non_blocking=True)
        self.copy_stream.synchronize()

    def _submit(self, slot: int, batch: dict) -> dict:
        if self._used[slot]:
            self.consumed[slot].synchronize()          # host buffer and device buffer of this slot are free again
        view = {}
        for k, v in batch.items():
            src = v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            n = src.shape[0]
            # pageable -> pinned with ONE thread (numpy): torch's copy_ fans a 34 MB copy out over every core, and its
            # spinning OpenMP workers then starve the HIP runtime's completion thread (measured: 30 ms steps instead of 9)
            np.copyto(self.host_np[slot][k][:n], src)
            view[k] = n
        with torch.cuda.stream(self.copy_stream):
            for k, n in view.items():
                self.dev[slot][k][:n].copy_(self.host[slot][k][:n], non_blocking=True)
            self.copied[slot].record(self.copy_stream)
        self._used[slot] = True
        return {k: self.dev[slot][k][:n] for k, n in view.items()}

    class Batch(dict):
        """Device views of one fed batch.  By default its slot is considered consumed by whatever the CURRENT stream has
        enqueued when the consumer asks for the next batch; a consumer that works on another stream (e.g. a
        models.PMCE.Pipeline lane) calls ``release(event)`` with an event recorded after its last use instead."""

        def __init__(self, views, feeder, slot):
            super().__init__(views)
            self._feeder, self._slot, self._released = feeder, slot, False

        def release(self, event: "torch.cuda.Event"):
            self._feeder.consumed[self._slot] = event
            self._released = True

    def run(self, host_batches):
        it = iter(host_batches)
        pending = []                                     # (slot, device views), oldest first
        slot = 0
        for _ in range(self.slots - 1):                  # prefill
            b = next(it, None)
            if b is None:
                break
            pending.append((slot, self._submit(slot, b)))
            slot = (slot + 1) % self.slots
        while pending:
            s, views = pending.pop(0)
            torch.cuda.current_stream(self.device).wait_event(self.copied[s])
            batch = PinnedFeeder.Batch(views, self, s)
            yield batch                                  # the consumer enqueues its kernels for this batch ...
            if not batch._released:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.device))
                self.consumed[s] = ev
            b = next(it, None)                           # ... and only then is the next slot refilled: the wait in _submit
            if b is not None:                            # is for the batch BEFORE the one just enqueued, so the GPU
                pending.append((slot, self._submit(slot, b)))   # always has work queued while the host copies
                slot = (slot + 1) % self.slots
