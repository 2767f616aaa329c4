"""Temporal pose encoder façade — drop-in for reference lib/models/PoseEstimation.py (GraphormerNet)."""
from __future__ import annotations

import ctypes as C

import torch

from .. import _lib, packing, synth
from ..config import FEAT_DIM, SEQLEN, cfg
from ..runtime import HipEngine, HipModuleBase, _check_input, build_param_tree


class GraphormerNet(HipModuleBase):
    """forward(x[B,16,J,2], img_feat[B,16,2048]) -> pose3d[B,J,3] in millimetres (PoseEstimation.py:95-115)."""

    def __init__(self, num_frames=16, num_joints=17, embed_dim=256, depth=3, pretrained=False, allow_pickle=None):
        super().__init__()
        if num_frames != SEQLEN:
            raise ValueError("the path is specialised for 16-frame clips (cfg.DATASET.seqlen, config.py:48)")
        self.num_joints, self.embed_dim, self.depth = num_joints, embed_dim, depth
        build_param_tree(self, synth.lifter_spec(num_joints, embed_dim, depth))
        self.eval()
        if pretrained:   # PoseEstimation.py:71-74
            from ..checkpoint import torch_load_checkpoint
            # restricted unpickler: tensors + what the reference's training loop logs (numpy scalars, the scheduler's Counter);
            # allow_pickle (argument, or cfg.MODEL.posenet_allow_pickle) opts into the unrestricted one for other files
            ap = getattr(cfg.MODEL, "posenet_allow_pickle", False) if allow_pickle is None else allow_pickle
            ckpt = torch_load_checkpoint(cfg.MODEL.posenet_path, map_location="cpu", allow_pickle=ap)
            self.load_state_dict(ckpt["model_state_dict"])

    def _build_engine(self, dev):
        eng = HipEngine(self.num_joints, self.embed_dim, self.depth)
        eng.register(packing.pack_lifter(self.state_dict(), "", dev, self.num_joints, self.embed_dim, self.depth))
        eng.finalize()
        return eng

    @torch.no_grad()
    def forward(self, x, img_feat):
        """Under the module's overflow policy (default "rerun": HipModuleBase.set_overflow_policy)."""
        x = _check_input(x, (SEQLEN, self.num_joints, 2), "pose2d")
        img_feat = _check_input(img_feat, (SEQLEN, FEAT_DIM), "img_feat")
        B = x.shape[0]
        if B == 0:
            return torch.empty(0, self.num_joints, 3, device=x.device, dtype=torch.float32)

        def launch(eng):
            out = torch.empty(B, self.num_joints, 3, device=x.device, dtype=torch.float32)
            ws = eng.workspace(B)
            _lib.check(eng.lib.pmce_lifter_forward(eng.handle, _lib.ptr(x), _lib.ptr(img_feat), _lib.ptr(out), B,
                                                   C.c_void_p(ws.data_ptr()), ws.numel(), _lib.current_stream()),
                       "pmce_lifter_forward")
            return out

        return self._guarded(launch)


def get_model(num_joint=17, embed_dim=256, depth=3, pretrained=False, allow_pickle=None):
    """Same signature as reference PoseEstimation.get_model (PoseEstimation.py:118-120) (+ allow_pickle, see GraphormerNet)."""
    return GraphormerNet(num_frames=cfg.DATASET.seqlen, num_joints=num_joint, embed_dim=embed_dim, depth=depth,
                         pretrained=pretrained, allow_pickle=allow_pickle)
