"""Host-side runtime shared by the three façade modules: a parameter tree with the reference's state_dict
keys, lazy packing onto the device, the ``pmce_model`` handle of libpmce_hip.so and its workspace.

PyTorch is plumbing here (device memory, streams, state_dict I/O); every FLOP of the forward runs in
libpmce_hip.so.  There is no CPU or eager fallback: calling forward on CPU tensors raises.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from . import _lib, assets, packing
from .config import NUM_VERTS, NUM_VERTS_FULL, SEQLEN, FEAT_DIM


class _Node(nn.Module):
    """Anonymous container; exists only so that state_dict keys equal the reference's."""


def build_param_tree(root: nn.Module, spec: "OrderedDict[str, tuple]", buffers=("init_vertices",)):
    """Create nested sub-modules/parameters so that ``root.state_dict().keys() == spec.keys()``."""
    for key, entry in spec.items():
        shape = entry[0] if isinstance(entry[0], (tuple, list)) else entry   # synth specs are (shape, half, off)
        parts = key.split(".")
        mod = root
        for p in parts[:-1]:
            if p not in mod._modules:
                mod.add_module(p, _Node())
            mod = mod._modules[p]
        t = torch.zeros(shape, dtype=torch.float32)
        if parts[-1] in buffers:
            mod.register_buffer(parts[-1], t)
        else:
            mod.register_parameter(parts[-1], nn.Parameter(t, requires_grad=False))


class HipEngine:
    """Owns the C handle, the packed tensors and the workspace for one module instance."""

    def __init__(self, num_joint: int, embed_dim: int, depth: int):
        self.lib = _lib.load()
        self.J, self.C, self.depth = num_joint, embed_dim, depth
        h = C.c_void_p()
        _lib.check(self.lib.pmce_model_create(num_joint, embed_dim, depth, C.byref(h)), "pmce_model_create")
        self.handle = h
        self.packed = OrderedDict()
        self.ws = None
        self.ws_batch = 0
        self.device = None
        self.regressor_rows = 0
        self.split_arena = None      # torch memory holding the f16 planes (kept alive here; lanes keep the owner's)
        self._finalized = False

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.pmce_model_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def tensor_names(self):
        n = self.lib.pmce_model_tensor_count(self.handle)
        return [self.lib.pmce_model_tensor_name(self.handle, i).decode() for i in range(n)]

    def register(self, tensors: "OrderedDict[str, torch.Tensor]"):
        for name, t in tensors.items():
            assert t.is_cuda and t.is_contiguous(), name
            self.packed[name] = t      # keep alive
            _lib.check(self.lib.pmce_model_set_tensor(self.handle, name.encode(), C.c_void_p(t.data_ptr())),
                       f"set_tensor({name})")
        self.device = next(iter(tensors.values())).device

    def set_regressor(self, j_regressor):
        csr, rows = packing.pack_regressor(j_regressor, self.device)
        self.register(csr)
        _lib.check(self.lib.pmce_model_set_regressor_rows(self.handle, rows), "set_regressor_rows")
        self.regressor_rows = rows

    def _give_split_arena(self):
        """The f16 planes of the split mode live in memory of PyTorch's allocator (the library allocates none of its own)."""
        n = int(self.lib.pmce_model_split_bytes(self.handle))
        if n and self.device is not None:
            self.split_arena = torch.empty(n, dtype=torch.uint8, device=self.device)
            _lib.check(self.lib.pmce_model_set_split_arena(self.handle, C.c_void_p(self.split_arena.data_ptr()), n), "model_set_split_arena")

    def finalize(self):
        if self.gemm_mode() == "split_f16":
            self._give_split_arena()
        # on the CURRENT stream: the packed fp32 tensors registered above were produced on it, and the library's own packing of the
        # f16 planes reads them (a torch side stream is non-blocking: the null stream is not ordered behind it)
        _lib.check(self.lib.pmce_model_finalize_on(self.handle, _lib.current_stream()), "pmce_model_finalize")
        self._finalized = True

    def clone_shared(self) -> "HipEngine":
        """A second handle on the SAME packed weights (no copy): its own workspace, side stream and events, so that two
        forwards can be in flight at once (models.PMCE.Pipeline).  The lane inherits the owner's overflow policy and small-batch
        threshold (pmce_model_share_split_weights copies both, whatever set them - API or environment)."""
        other = HipEngine(self.J, self.C, self.depth)
        other.set_gemm_mode(self.gemm_mode())
        other.register(self.packed)
        _lib.check(other.lib.pmce_model_share_split_weights(other.handle, self.handle), "model_share_split_weights")
        if self.gemm_mode() != "split_f16":     # (nothing to share on the fp32 pipe: copy the two settings by hand)
            other.set_split_min_batch(self.get_split_min_batch())
            other.set_overflow_policy(self.strict_overflow)
        if self.regressor_rows:
            _lib.check(other.lib.pmce_model_set_regressor_rows(other.handle, self.regressor_rows), "set_regressor_rows")
            other.regressor_rows = self.regressor_rows
        other.split_arena = self.split_arena      # the planes it adopts live in this engine's arena
        other.finalize()
        return other

    def workspace(self, batch: int):
        # compared in BYTES, not batch counts: the library's answer is what check_ws enforces
        nbytes = self.lib.pmce_model_workspace_bytes(self.handle, batch)
        if self.ws is None or nbytes > self.ws.numel() or self.ws.device != self.device:
            if self.ws is not None and self.ws.is_cuda:
                # forwards enqueued on this stream may still be using the old buffer: the allocator must not hand it out
                # to another stream before they are done
                self.ws.record_stream(torch.cuda.current_stream(self.ws.device))
            self.ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self.ws_batch = batch
        return self.ws

    def intermediate(self, name: str, batch: int, shape):
        """View of a named workspace buffer after a forward of ``batch`` clips (tests only)."""
        off = self.lib.pmce_model_workspace_offset(self.handle, batch, name.encode())
        if off < 0:
            raise KeyError(name)
        n = int(np.prod(shape))
        return self.ws[off:off + 4 * n].view(torch.float32).reshape(shape)

    def set_gemm_mode(self, mode: str):
        """'split_f16' (default): the lifter's Linear layers on the f16 matrix pipe in the three-product form (fp32 operands,
        fp32 accumulate, fp32 accuracy); 'f32': on the fp32 matrix pipe."""
        if mode not in ("split_f16", "f32"):
            raise ValueError("gemm mode must be 'split_f16' or 'f32'")
        if mode == "split_f16" and self.gemm_mode() != "split_f16" and self._finalized:
            self._give_split_arena()      # a finalized fp32-mode model is about to pack its planes
        _lib.check(self.lib.pmce_model_set_gemm_mode_on(self.handle, 1 if mode == "split_f16" else 0, _lib.current_stream()),
                   "model_set_gemm_mode")
        if mode == "f32":
            self.split_arena = None       # (the library waited for the device before it let go of the planes)

    def gemm_mode(self) -> str:
        return "split_f16" if self.lib.pmce_model_gemm_mode(self.handle) else "f32"

    def set_split_min_batch(self, clips: int):
        _lib.check(self.lib.pmce_model_set_split_min_batch(self.handle, int(clips)), "model_set_split_min_batch")

    def get_split_min_batch(self) -> int:
        """The threshold in force (its default comes from PMCE_SPLIT_MIN_BATCH at create time)."""
        return int(self.lib.pmce_model_get_split_min_batch(self.handle))

    def overflowed(self) -> bool:
        """A product of the split-f16 form produced a non-finite value in a completed call (non-finite inputs, an intermediate
        activation beyond f16's 65504 under exotic weights, or fp32 overflow): the affected clips' outputs are inf / nan.  The word
        only REPORTS (sticky until :meth:`clear_overflow`); with :meth:`set_overflow_policy` ``strict=True`` further calls raise
        instead.  Synchronise the stream first for a definite answer about the last call."""
        return bool(self.lib.pmce_model_overflowed(self.handle))

    def set_overflow_policy(self, strict: bool):
        _lib.check(self.lib.pmce_model_set_overflow_policy(self.handle, 1 if strict else 0), "model_set_overflow_policy")

    @property
    def strict_overflow(self) -> bool:
        """The C-side policy in force (its default comes from PMCE_STRICT_OVERFLOW at create time)."""
        return self.lib.pmce_model_get_overflow_policy(self.handle) == 1

    def clear_overflow(self):
        _lib.check(self.lib.pmce_model_clear_overflow(self.handle), "model_clear_overflow")

    def run_on_f32_pipe(self, launch):
        """launch(self) with every product on the fp32 matrix pipe - the reference's own range - while the f16 planes stay packed: the
        threshold below which calls stay on the fp32 pipe is raised above any batch for the duration of the call and RESTORED to the
        value that was in force (also when that came from the environment)."""
        prev = self.get_split_min_batch()
        self.set_split_min_batch(1 << 30)
        try:
            return launch(self)
        finally:
            self.set_split_min_batch(prev)

    def set_concurrency(self, enable: bool):
        _lib.check(self.lib.pmce_model_set_concurrency(self.handle, 1 if enable else 0), "model_set_concurrency")

    # ---- profiling -------------------------------------------------------------------------------
    def profile(self, enable: bool):
        _lib.check(self.lib.pmce_model_profile(self.handle, 1 if enable else 0), "model_profile")

    def profile_read(self):
        name, ms, n = C.c_char_p(), C.c_double(), C.c_longlong()
        count = self.lib.pmce_model_profile_read(self.handle, -1, None, None, None)
        out = OrderedDict()
        for i in range(count):
            self.lib.pmce_model_profile_read(self.handle, i, C.byref(name), C.byref(ms), C.byref(n))
            out[name.value.decode()] = (ms.value, n.value)
        return out


def _check_input(x, shape_tail, name):
    if not x.is_cuda:
        raise _lib.PmceError(f"{name} is on {x.device}: the HIP path needs GPU tensors and there is no CPU fallback")
    if tuple(x.shape[1:]) != tuple(shape_tail):
        raise ValueError(f"{name} must be [B,{','.join(map(str, shape_tail))}], got {tuple(x.shape)}")
    return x.to(torch.float32).contiguous()


class HipModuleBase(nn.Module):
    """Common behaviour of the façades: reference-layout parameters, lazy (re)packing, inference only."""

    def __init__(self):
        super().__init__()
        self._engine = None
        self._dirty = True
        self._gemm_mode = None   # None = the library's default (split_f16 unless PMCE_SPLIT_F16=0)
        self._split_min_batch = None
        self._overflow_policy = None   # None = "rerun" (or "strict" under PMCE_STRICT_OVERFLOW=1): see set_overflow_policy
        self.overflow_reruns = 0       # calls that were computed again on the fp32 pipe under the "rerun" policy
        self._overflow_carried = False  # a report of EARLIER asynchronous work that a "rerun" call took out of the shared word (see _guarded)

    def set_gemm_mode(self, mode, min_batch=None):
        """Arithmetic of the large products (min_batch: calls with fewer clips stay on the fp32 pipe, default 1 = none): 'split_f16' (three f16 products per fp32 product on the f16 matrix
        pipe, fp32 accumulate; fp32-grade accuracy, the default) or 'f32' (fp32 matrix pipe).  Takes effect at the next
        forward; pipelines and captured graphs made before must be rebuilt.

        On MI355X a wave executing f16 matrix instructions corrupts packed-fp32 arithmetic (v_pk_{fma,mul,add}_f32) of other waves
        on the same CU (scripts/microbench/victims.py).  The library is therefore built without any packed-fp32 instruction
        (build.py; checked on the device code by tests/test_host_logic.py), which makes its own kernels safe next to each other:
        both modes use the two-stream / two-batches-in-flight execution, and overlapped runs are bitwise equal to serial ones
        (tests/test_gpu_e2e.py::test_overlapped_split_mode_equals_serial).  OTHER GPU work of the process: what the suite measures
        on every run (tests/test_gpu_bystander.py) is that torch's elementwise / normalisation / reduction kernels and
        compiler-generated packed fp32 stay bit-correct beside forwards in both modes; the affected form is hand-written
        v_pk_fma_f32 with op_sel operands - code like that should not run concurrently with a forward in 'split_f16' mode
        ('f32' mode has no such restriction; PMCE_SPLIT_OVERLAP=0 restores the strictly serial schedule)."""
        if mode not in (None, "split_f16", "f32"):
            raise ValueError("gemm mode must be 'split_f16', 'f32' or None")
        self._gemm_mode = mode
        self._split_min_batch = min_batch
        self._dirty = True

    def gemm_mode(self):
        return self._ensure_packed().gemm_mode()

    def overflowed(self, synchronize: bool = True) -> bool:
        """True if a product of a call on this module produced a non-finite value: non-finite INPUTS (they propagate into the
        outputs of their own clip, as in the reference), an intermediate activation beyond f16's range under exotic weights, or
        fp32 overflow.  Finite inputs of any magnitude are in range (img_feat is row-scaled before its products).  Under the default
        policy ("rerun") ``forward`` has already dealt with it and cleared the word; under "report" the affected clips' outputs are
        inf / nan, other clips and later calls are unaffected; see :meth:`set_overflow_policy`."""
        eng = self._ensure_packed()
        if synchronize:
            torch.cuda.synchronize(eng.device)
        return eng.overflowed() or self._overflow_carried

    OVERFLOW_POLICIES = ("rerun", "report", "strict")

    def set_overflow_policy(self, policy="rerun"):
        """What ``forward`` (and ``forward_with_joints``) do about the one error mode the split-f16 arithmetic has and the reference
        lacks - an intermediate activation beyond f16's 65504, possible only with weights that drive a LayerNorm / attention / GELU /
        GRU output there:

        * ``"rerun"`` (default): the call waits for its result on the current stream, polls the model's overflow word and, if a product
          reported a non-finite value, computes the batch again on the fp32 matrix pipe (the reference's own range) and clears the
          word: the caller of ``model(pose2d, img_feat)`` (lib/core/base.py:222) never sees a value the reference would not have
          produced.  (Non-finite INPUTS still give non-finite outputs for their own clips, as in the reference.)  The wait costs the
          launch-ahead of back-to-back calls; throughput loops use ``Pipeline`` or ``"report"``.
        * ``"report"``: fully asynchronous calls; the word only reports (``overflowed()``, ``Pipeline.synchronize``, the evaluator).
        * ``"strict"``: as "report", and while the word is set every further call raises PmceError until :meth:`clear_overflow`.

        Booleans are accepted for the round-4 meaning (True = "strict", False = "report")."""
        if policy is True:
            policy = "strict"
        elif policy is False:
            policy = "report"
        if policy not in self.OVERFLOW_POLICIES:
            raise ValueError(f"overflow policy must be one of {self.OVERFLOW_POLICIES}")
        self._overflow_policy = policy
        self._ensure_packed().set_overflow_policy(policy == "strict")

    def overflow_policy(self) -> str:
        if self._overflow_policy is None:        # default: the environment's strict switch, else "rerun"
            import os
            self._overflow_policy = "strict" if os.environ.get("PMCE_STRICT_OVERFLOW", "0") not in ("", "0") else "rerun"
        return self._overflow_policy

    def clear_overflow(self):
        self._overflow_carried = False
        self._ensure_packed().clear_overflow()

    def _guarded(self, launch, eng=None):
        """launch(engine) -> outputs, under the module's overflow policy (see set_overflow_policy).

        The overflow word is shared by every engine on these weights (pipeline lanes) and sticky.  A word that is ALREADY set when a
        "rerun" call starts belongs to earlier asynchronous work (a "report"-mode call, a lane): it must neither trigger a re-run of
        THIS batch on the fp32 pipe (its numbers would then depend on what ran before it) nor be erased unseen.  It is moved into
        ``_overflow_carried`` - still reported by :meth:`overflowed` and ``Pipeline.synchronize`` until :meth:`clear_overflow` - and the
        word is cleared before the launch, so that only this call's products can set it."""
        eng = eng or self._ensure_packed()
        rerun = self.overflow_policy() == "rerun" and eng.gemm_mode() == "split_f16" and not torch.cuda.is_current_stream_capturing()
        if rerun and eng.overflowed():
            torch.cuda.current_stream(eng.device).synchronize()
            self._overflow_carried = True
            eng.clear_overflow()
        out = launch(eng)
        if not rerun:
            return out
        st = torch.cuda.current_stream(eng.device)
        st.synchronize()
        if not eng.overflowed():
            return out
        eng.clear_overflow()
        out = eng.run_on_f32_pipe(launch)
        st.synchronize()
        self.overflow_reruns += 1
        return out

    # any change of the parameters' storage invalidates the packed copy
    def _apply(self, fn, *a, **k):
        self._dirty = True
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, state_dict, strict=True, **kw):
        """Accepts the reference's checkpoint layout, with or without the checkpoint wrapper / 'module.' prefix."""
        sd = packing.unwrap_checkpoint(state_dict)
        self._dirty = True
        return super().load_state_dict(sd, strict=strict, **kw)

    def train(self, mode: bool = True):
        if mode:
            raise NotImplementedError("pmce_amd implements the inference hot path only (training is out of scope)")
        return super().train(False)

    def _device(self):
        return next(self.parameters()).device

    def _ensure_packed(self):
        dev = self._device()
        if dev.type != "cuda":
            raise _lib.PmceError("model parameters are on CPU: move the model to the GPU (.cuda()); no CPU fallback exists")
        if self._engine is None or self._dirty or self._engine.device != dev:
            self._engine = self._build_engine(dev)
            if self._gemm_mode is not None:
                self._engine.set_gemm_mode(self._gemm_mode)
            if self._split_min_batch is not None:
                self._engine.set_split_min_batch(self._split_min_batch)
            if self._overflow_policy is not None:
                self._engine.set_overflow_policy(self._overflow_policy == "strict")
            self._dirty = False
        return self._engine

    def _build_engine(self, dev):  # pragma: no cover - abstract
        raise NotImplementedError
