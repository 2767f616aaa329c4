"""Pose–mesh co-evolution decoder façade — drop-in for reference lib/models/CoevoDecoder.py (Pose2Mesh)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _lib, assets, packing, synth
from ..config import FEAT_DIM, NUM_VERTS_FULL, SEQLEN, cfg
from ..runtime import HipEngine, HipModuleBase, _check_input, build_param_tree


class Pose2Mesh(HipModuleBase):
    """forward(joints[B,J,3] (m), img_feats[B,16,2048]) -> (joints3[B,J,3], mesh[B,6890,3]) (CoevoDecoder.py:226-246).

    Init-time work of the reference (CoevoDecoder.py:197-209) happens here on the host: the 431-vertex template
    (buffer ``init_vertices``) and the per-vertex nearest-joint table ``vj_relation``."""

    def __init__(self, num_joint, embed_dim=256, base_data_dir=None):
        super().__init__()
        self.num_joint = num_joint
        build_param_tree(self, synth.decoder_spec(num_joint, cfg.MODEL.joint_dim))
        v431, vj, src = assets.build_template(base_data_dir)
        self.init_vertices.copy_(torch.from_numpy(v431))
        self.vj_relation = vj            # int64[431], values in 0..16
        self.num_verts = v431.shape[0]
        self.base_data_source = src
        self.eval()

    def _build_engine(self, dev):
        eng = HipEngine(self.num_joint, 256, 3)
        eng.register(packing.pack_decoder(self.state_dict(), "", dev, self.num_joint, self.vj_relation))
        eng.finalize()
        return eng

    @torch.no_grad()
    def forward(self, joints, img_feats):
        """Under the module's overflow policy (default "rerun": HipModuleBase.set_overflow_policy)."""
        joints = _check_input(joints, (self.num_joint, 3), "joints")
        img_feats = _check_input(img_feats, (SEQLEN, FEAT_DIM), "img_feats")
        B = joints.shape[0]
        if B == 0:
            return (torch.empty(0, self.num_joint, 3, device=joints.device, dtype=torch.float32),
                    torch.empty(0, NUM_VERTS_FULL, 3, device=joints.device, dtype=torch.float32))

        def launch(eng):
            pose = torch.empty(B, self.num_joint, 3, device=joints.device, dtype=torch.float32)
            mesh = torch.empty(B, NUM_VERTS_FULL, 3, device=joints.device, dtype=torch.float32)
            ws = eng.workspace(B)
            _lib.check(eng.lib.pmce_decoder_forward(eng.handle, _lib.ptr(joints), _lib.ptr(img_feats), _lib.ptr(pose),
                                                    _lib.ptr(mesh), B, C.c_void_p(ws.data_ptr()), ws.numel(),
                                                    _lib.current_stream()), "pmce_decoder_forward")
            return pose, mesh

        return self._guarded(launch)


def get_model(num_joint, embed_dim):
    """Same signature as reference CoevoDecoder.get_model (CoevoDecoder.py:249-252)."""
    return Pose2Mesh(num_joint, embed_dim)
