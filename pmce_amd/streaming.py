"""Sliding-window ("streaming") evaluation of long sequences — BASELINE config 5, SURVEY §8f ranks 2-3.

The reference handles a video by cutting it into stride-1 windows of 16 frames, each an independent clip that predicts its
middle frame (lib/_img_utils.py:27-57 ``split_into_chunks_pose``; demo: lib/utils/_dataset_demo.py:91-104 with the first and
last 8 frames served by a single frame repeated 16 times).  Every frame's 2048-d feature would be copied 16 times if the
windows were materialised on the host; here the per-frame tables are uploaded once and the windows are assembled on the
GPU by ``pmce_assemble_windows_f32`` (16-byte vector copies, HBM-bound).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .config import FEAT_DIM, SEQLEN


def window_indices(num_frames: int, seqlen: int = SEQLEN, stride: int = 1, match_vibe: bool = True) -> np.ndarray:
    """[start, end] (inclusive) of every window of ONE video, as ``split_into_chunks_pose`` produces them
    (lib/_img_utils.py:42-55): all stride-``stride`` windows, then — when stride != seqlen and match_vibe — trailing windows
    are dropped so that the last window ends where the last full 16-frame VIBE chunk ends."""
    if num_frames < seqlen:
        return np.zeros((0, 2), dtype=np.int64)
    starts = np.arange(0, num_frames - seqlen + 1, stride)
    sf = np.stack([starts, starts + seqlen - 1], 1)
    if stride != seqlen and match_vibe:
        vibe_last_end = (num_frames // 16) * 16 - 1          # last element of view_as_windows(indexes, 16, step=16)[-1]
        for j in range(1, len(sf) + 1):
            if sf[-j][1] == vibe_last_end:
                if j != 1:
                    sf = sf[: len(sf) - j + 1]
                break
    return sf.astype(np.int64)


def demo_window_list(num_frames: int, seqlen: int = SEQLEN) -> np.ndarray:
    """One window per frame as the demo builds them (lib/utils/_dataset_demo.py:91-96): frames 0..seqlen/2-1 and the last
    seqlen/2-1 frames use a single frame repeated seqlen times ([i,i]), the others the window whose middle they are."""
    if num_frames < seqlen:
        # the reference's list (head 0..7, tail -7..-1) is not one window per frame below seqlen frames: frames repeat or
        # index out of range.  Refuse instead of returning wrong clips.
        raise ValueError(f"demo_window_list needs at least {seqlen} frames (got {num_frames})")
    h = seqlen // 2
    mid = [[i, i + seqlen - 1] for i in range(num_frames - seqlen + 1)]
    head = [[i, i] for i in range(h)]
    tail = [[num_frames - h + i, num_frames - h + i] for i in range(1, h)]
    return np.asarray(head + mid + tail, dtype=np.int64)


def validate_windows(windows, num_frames: int, seqlen: int = SEQLEN) -> np.ndarray:
    """int32[W,2] window table checked on the host before any launch: 0 <= start <= end < num_frames and a window is
    either ``seqlen`` consecutive frames or one repeated frame (start == end).  The kernels clamp frame indices, so an
    out-of-range table (e.g. ``demo_window_list`` of a sequence shorter than ``seqlen``) would otherwise yield wrong clips
    silently."""
    w = np.asarray(windows)
    if w.size == 0:
        return np.zeros((0, 2), dtype=np.int32)
    if w.ndim != 2 or w.shape[1] != 2:
        raise ValueError(f"windows must be [W,2] (start, end inclusive), got shape {w.shape}")
    s, e = w[:, 0].astype(np.int64), w[:, 1].astype(np.int64)
    bad = (s < 0) | (e >= num_frames) | (s > e) | ((e - s != 0) & (e - s != seqlen - 1))
    if bad.any():
        i = int(np.argmax(bad))
        raise ValueError(f"window {i} = [{int(s[i])}, {int(e[i])}] is invalid for a sequence of {num_frames} frames "
                         f"(need 0 <= start <= end < {num_frames} and end - start in {{0, {seqlen - 1}}})")
    return np.ascontiguousarray(w, dtype=np.int32)


def assemble_windows(pose2d_frames: torch.Tensor, feat_frames: torch.Tensor, windows) -> tuple:
    """Per-frame tables pose2d[L,J,2], feat[L,2048] (GPU) + windows int[W,2] -> (pose2d[W,16,J,2], img_feat[W,16,2048]).
    start == end means "this frame repeated 16 times" (the demo's head/tail windows)."""
    lib = _lib.load()
    dev = feat_frames.device
    p = pose2d_frames.to(torch.float32).contiguous()
    f = feat_frames.to(torch.float32).contiguous()
    L, J, _ = p.shape
    w = torch.as_tensor(validate_windows(windows, L), device=dev).contiguous()
    W = w.shape[0]
    out_p = torch.empty(W, SEQLEN, J, 2, device=dev, dtype=torch.float32)
    out_f = torch.empty(W, SEQLEN, FEAT_DIM, device=dev, dtype=torch.float32)
    _lib.check(lib.pmce_assemble_windows_f32(_lib.ptr(p), _lib.ptr(f), _lib.ptr(w), _lib.ptr(out_p), _lib.ptr(out_f), W, L, J,
                                             _lib.current_stream()), "assemble_windows")
    return out_p, out_f


class FrameCache:
    """Per-frame tables of one sequence for the reuse path: x0[L,J,C] (lifter tokens after the window-independent first
    spatial block) and gi0[L,6144] (GRU layer-0 input projections).  Built once by :func:`precompute_frames`."""

    def __init__(self, x0, gi0, num_frames):
        self.x0, self.gi0, self.L = x0, gi0, num_frames


@torch.no_grad()
def precompute_frames(model, pose2d_frames, feat_frames) -> FrameCache:
    import ctypes as C
    eng = model._ensure_packed()
    p = pose2d_frames.to(torch.float32).contiguous()
    f = feat_frames.to(torch.float32).contiguous()
    L, J, _ = p.shape
    dev = f.device
    x0 = torch.empty(L, J, model.embed_dim, device=dev, dtype=torch.float32)
    gi0 = torch.empty(L, 6144, device=dev, dtype=torch.float32)
    ws = eng.workspace((L + SEQLEN - 1) // SEQLEN)
    _lib.check(eng.lib.pmce_stream_precompute(eng.handle, _lib.ptr(p), _lib.ptr(f), L, _lib.ptr(x0), _lib.ptr(gi0),
                                              C.c_void_p(ws.data_ptr()), ws.numel(), _lib.current_stream()), "stream_precompute")
    return FrameCache(x0, gi0, L)


@torch.no_grad()
def stream_forward_cached(model, cache: FrameCache, windows=None, batch: int = 256, with_joints: bool = False, lanes: int = 2):
    """Same outputs as :func:`stream_forward`, but the per-frame work is taken from ``cache`` (about 21 % fewer FLOPs per
    window: SpatialBlocks[0], imgfeat_embed and the GRU layer-0 input projection are not recomputed 16 times per frame).
    ``lanes`` window batches are in flight at once on handles that share the packed weights (models.PMCE.Pipeline's
    scheme: batch k on lane k % lanes, its lifter started when the previous batch's has finished); results do not depend
    on it."""
    import ctypes as C
    from .config import NUM_VERTS_FULL
    main = model._ensure_packed()
    windows = validate_windows(window_indices(cache.L) if windows is None else windows, cache.L)
    dev = cache.x0.device
    J = model.num_joint
    nb = (len(windows) + batch - 1) // batch
    lanes = max(1, min(lanes, nb))
    if getattr(model, "_stream_lanes", None) is None or len(model._stream_lanes[0]) < lanes or model._stream_lanes[0][0] is not main:
        if main.gemm_mode() == "split_f16" and not _lib.split_overlap():   # diagnostic: one stream for every lane
            sts = [torch.cuda.Stream(device=dev)] * lanes
        else:
            sts = [torch.cuda.Stream(device=dev) for _ in range(lanes)]
        model._stream_lanes = ([main] + [main.clone_shared() for _ in range(lanes - 1)], sts)
    engines, streams = model._stream_lanes
    cur = torch.cuda.current_stream(dev)
    ready = torch.cuda.Event()
    ready.record(cur)
    outs, prev = [], None
    for k, lo in enumerate(range(0, len(windows), batch)):
        eng, st = (engines[k % lanes], streams[k % lanes]) if lanes > 1 else (main, cur)
        w = torch.as_tensor(np.asarray(windows[lo:lo + batch], dtype=np.int32), device=dev).contiguous()
        W = w.shape[0]
        if lanes > 1:
            st.wait_event(ready)
            if prev is not None and prev is not eng:
                _lib.check(eng.lib.pmce_model_wait_lifter(prev.handle, C.c_void_p(st.cuda_stream)), "model_wait_lifter")
        prev = eng
        with torch.cuda.stream(st):
            mesh = torch.empty(W, NUM_VERTS_FULL, 3, device=dev, dtype=torch.float32)
            pose = torch.empty(W, J, 3, device=dev, dtype=torch.float32)
            pose3d = torch.empty(W, J, 3, device=dev, dtype=torch.float32)
            pred = torch.empty(W, eng.regressor_rows, 3, device=dev, dtype=torch.float32) if with_joints else None
            ws = eng.workspace(max(W, batch))
            _lib.check(eng.lib.pmce_stream_forward(eng.handle, _lib.ptr(cache.x0), _lib.ptr(cache.gi0), _lib.ptr(w), W, cache.L,
                                                   _lib.ptr(mesh), _lib.ptr(pose), _lib.ptr(pose3d), _lib.ptr(pred),
                                                   C.c_void_p(ws.data_ptr()), ws.numel(), C.c_void_p(st.cuda_stream)),
                       "stream_forward")
            if lanes > 1:
                w.record_stream(st)
        outs.append((mesh, pose, pose3d, pred) if with_joints else (mesh, pose, pose3d))
    if lanes > 1:
        for st in streams[:lanes]:                       # join: the caller's stream continues after every lane
            done = torch.cuda.Event()
            done.record(st)
            cur.wait_event(done)
        for o in outs:
            for t in o:
                if t is not None:
                    t.record_stream(cur)
    return tuple(torch.cat([o[i] for o in outs], 0) for i in range(len(outs[0]))) if outs else ()


@torch.no_grad()
def stream_forward(model, pose2d_frames, feat_frames, windows=None, batch: int = 256, with_joints: bool = False):
    """Run ``model`` over all windows of one sequence in batches of ``batch`` clips; returns the concatenated outputs
    (one row per window = per predicted middle frame)."""
    L = feat_frames.shape[0]
    windows = window_indices(L) if windows is None else np.asarray(windows)
    outs = []
    for lo in range(0, len(windows), batch):
        p, f = assemble_windows(pose2d_frames, feat_frames, windows[lo:lo + batch])
        outs.append(model.forward_with_joints(p, f) if with_joints else model(p, f))
    return tuple(torch.cat([o[i] for o in outs], 0) for i in range(len(outs[0]))) if outs else ()
