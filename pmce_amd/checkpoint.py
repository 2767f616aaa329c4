"""Checkpoint I/O for the hot path.

* :func:`load_reference_checkpoint` reads a reference ``*.pth.tar`` (``torch.save`` dict with ``'model_state_dict'``,
  main/train.py:57-64; lifter-only checkpoints have the same key, PoseEstimation.py:71-74), strips a DataParallel
  ``module.`` prefix (lib/funcs_utils.py:65-70), infers (num_joint, embed_dim, depth) from tensor shapes and validates every
  key/shape against the layout of SURVEY §8b, naming anything missing or unexpected instead of failing later on the GPU.
* :func:`export_packed` / :func:`load_packed` write/read the kernel-ready packed operands (pmce_amd.packing) as one
  safetensors file, so a serving process can mmap the weights without the reference layout or the load-time repacking.
"""
from __future__ import annotations

import re
from collections import OrderedDict

import numpy as np
import torch

from . import packing, synth


def infer_dims(sd):
    """(kind, num_joint, embed_dim, depth) from a state_dict; kind in {'pmce', 'lifter', 'decoder'}."""
    keys = list(sd.keys())
    if any(k.startswith("pose_lifter.") for k in keys):
        kind, lp = "pmce", "pose_lifter."
    elif "spatial_pos_embed" in sd:
        kind, lp = "lifter", ""
    elif any(k.startswith("coevoblock1.") for k in keys):
        j = sd["coevoblock1.joint_pos_embed"].shape[1]
        return "decoder", int(j), 256, 3
    else:
        raise ValueError("not a PMCE / GraphormerNet / Pose2Mesh state_dict")
    spe = sd[lp + "spatial_pos_embed"]
    depth = 1 + max(int(m.group(1)) for k in keys for m in [re.match(re.escape(lp) + r"SpatialBlocks\.(\d+)\.", k)] if m)
    return kind, int(spe.shape[1]), int(spe.shape[2]), depth


def validate_state_dict(sd):
    """Raise ValueError listing every missing / unexpected / mis-shaped tensor; returns (kind, J, C, depth)."""
    kind, J, C, depth = infer_dims(sd)
    spec = {"pmce": lambda: synth.pmce_spec(J, C, depth), "lifter": lambda: synth.lifter_spec(J, C, depth),
            "decoder": lambda: synth.decoder_spec(J)}[kind]()
    problems = []
    for k, (shape, _, _) in spec.items():
        if k not in sd:
            problems.append(f"missing: {k} {tuple(shape)}")
        elif tuple(sd[k].shape) != tuple(shape):
            problems.append(f"shape: {k} is {tuple(sd[k].shape)}, expected {tuple(shape)}")
    for k in sd:
        if k not in spec:
            problems.append(f"unexpected: {k}")
    if problems:
        raise ValueError(f"{kind} checkpoint (J={J}, C={C}, depth={depth}) does not match the reference layout:\n  "
                         + "\n  ".join(problems[:40]) + ("\n  ..." if len(problems) > 40 else ""))
    return kind, J, C, depth


def _reference_checkpoint_globals():
    """What a checkpoint written by the reference's training loop holds besides tensors and plain containers (main/train.py:57-64:
    ``epoch``, ``model_state_dict``, ``optim_state_dict`` (Adam), ``scheduler_state_dict`` (MultiStepLR: a ``collections.Counter``
    of milestones) and ``train_log`` / ``test_log`` lists whose entries are NUMPY SCALARS, ``np.power(...).mean()`` in
    compute_both_err): the constructors of numpy scalars and dtypes, nothing that can run code."""
    import collections
    allowed = [collections.Counter, collections.OrderedDict, np.dtype, np.ndarray]
    for modname in ("numpy._core.multiarray", "numpy.core.multiarray"):     # numpy 2.x / 1.x
        try:
            mod = __import__(modname, fromlist=["scalar"])
        except Exception:  # noqa: BLE001
            continue
        for fn in ("scalar", "_reconstruct"):
            if hasattr(mod, fn):
                allowed.append(getattr(mod, fn))
    for tname in ("float16", "float32", "float64", "int8", "int16", "int32", "int64", "uint8", "bool_"):
        t = getattr(np, tname, None)
        if t is not None:
            allowed.append(t)
            allowed.append(type(np.dtype(t)))      # numpy.dtypes.Float32DType ... (numpy >= 1.25 pickles the dtype's class)
    return allowed


_REF_GLOBALS = None     # built once: the allow-list does not change during a process


def _safe_globals_ctx():
    """Context manager that adds the allow-list to torch's restricted unpickler.  ``torch.serialization.safe_globals`` exists
    since torch 2.5; torch 2.4 has the process-wide ``add_safe_globals`` only; older releases have neither and cannot read a
    reference checkpoint with ``weights_only=True`` at all - that is said in so many words instead of surfacing as an
    unpickling failure of a perfectly good file."""
    global _REF_GLOBALS
    if _REF_GLOBALS is None:
        _REF_GLOBALS = _reference_checkpoint_globals()
    ser = torch.serialization
    if hasattr(ser, "safe_globals"):
        return ser.safe_globals(_REF_GLOBALS)
    if hasattr(ser, "add_safe_globals"):
        import contextlib
        ser.add_safe_globals(_REF_GLOBALS)
        return contextlib.nullcontext()
    raise RuntimeError(f"torch {torch.__version__} has no allow-list for the restricted unpickler (needs torch >= 2.4): a reference "
                       "checkpoint (numpy scalars in its logs, a Counter in its scheduler state) can only be read with "
                       "allow_pickle=True on this torch")


def torch_load_checkpoint(path, map_location="cpu", allow_pickle=False):
    """torch.load that cannot execute code: ``weights_only=True`` with an allow-list of exactly the non-tensor objects the
    reference's own checkpoints contain (numpy scalars in the error logs, the scheduler's ``Counter`` - see
    :func:`_reference_checkpoint_globals`; without it a genuine ``best.pth.tar`` is rejected).  ``allow_pickle=True`` is the
    explicit opt-in to the unrestricted unpickler, used ONLY when the restricted one refused the file's contents
    (``pickle.UnpicklingError``) or this torch has no allow-list - never to paper over an unrelated failure."""
    import pickle
    try:
        ctx = _safe_globals_ctx()
    except RuntimeError:
        if not allow_pickle:
            raise
        return torch.load(path, map_location=map_location, weights_only=False)
    try:
        with ctx:
            return torch.load(path, map_location=map_location, weights_only=True)
    except pickle.UnpicklingError:
        if not allow_pickle:
            raise
    return torch.load(path, map_location=map_location, weights_only=False)


def load_reference_checkpoint(path, map_location="cpu", allow_pickle=False):
    """-> (state_dict, kind, num_joint, embed_dim, depth).  A file that cannot be read raises ValueError whose message starts
    with "No checkpoint exists!" like the reference's load_checkpoint (lib/funcs_utils.py:122-128) - followed by what actually
    went wrong (missing file, or an object the restricted unpickler refuses: pass allow_pickle=True for such a file)."""
    try:
        obj = torch_load_checkpoint(path, map_location, allow_pickle)
    except Exception as e:  # noqa: BLE001 - same contract as the reference
        import pickle
        hint = "" if not isinstance(e, pickle.UnpicklingError) or allow_pickle else \
            "  (the file exists but holds objects outside the tensors / numpy scalars / Counter a reference checkpoint contains; " \
            "allow_pickle=True loads it with the unrestricted unpickler)"
        raise ValueError(f"No checkpoint exists!\n{type(e).__name__}: {e}{hint}") from e
    sd = packing.unwrap_checkpoint(obj)
    kind, J, C, depth = validate_state_dict(sd)
    return sd, kind, J, C, depth


def export_packed(model, path):
    """Write the packed operands of a pmce_amd.models.PMCE instance (on the GPU) to a safetensors file."""
    from safetensors.torch import save_file
    eng = model._ensure_packed()
    tensors = OrderedDict((k, v.detach().cpu().contiguous()) for k, v in eng.packed.items())
    meta = {"format": "pmce_amd.packed.v1", "num_joint": str(model.num_joint), "embed_dim": str(model.embed_dim),
            "depth": str(model.depth)}
    save_file(tensors, path, metadata=meta)
    return meta


def load_packed(path, device):
    """-> (tensors on device, meta).  Register them with HipEngine.register() / pmce_model_set_tensor."""
    from safetensors import safe_open
    out = OrderedDict()
    with safe_open(path, framework="pt", device=str(device)) as f:
        meta = f.metadata()
        for k in f.keys():
            out[k] = f.get_tensor(k).contiguous()
    if not meta or meta.get("format") != "pmce_amd.packed.v1":
        raise ValueError(f"{path} is not a pmce_amd packed-weights file")
    return out, meta
