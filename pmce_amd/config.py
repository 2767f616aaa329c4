"""Tiny configuration for the PMCE per-clip inference hot path.

The reference keeps a global EasyDict ``cfg`` (reference lib/core/config.py:16-97) whose import has
filesystem side effects (config.py:9-13,38).  The hot path reads six scalars from it
(CoevoDecoder.py:14,211,229; PMCE.py:12; PoseEstimation.py:119).  This module carries exactly
those, with the same names, and nothing else.
"""
from __future__ import annotations

import os
from types import SimpleNamespace

cfg = SimpleNamespace(
    DATASET=SimpleNamespace(
        seqlen=16,                      # reference config.py:48
        BASE_DATA_DIR=os.environ.get("PMCE_BASE_DATA_DIR", "data/base_data"),  # config.py:53
        target_joint_set="human36",     # config/test_mesh_3dpw.yml:4 (selects the caller's J_regressor)
    ),
    MODEL=SimpleNamespace(
        hpe_dim=256,                    # config.py:59
        hpe_dep=3,                      # config.py:60
        joint_dim=64,                   # config.py (MODEL.joint_dim)
        vertx_dim=64,                   # config.py (MODEL.vertx_dim)
        posenet_pretrained=False,
        posenet_path="",
        posenet_allow_pickle=False,     # (not in the reference) unrestricted unpickler for GraphormerNet(pretrained=True)
    ),
)

# Fixed structural constants of the path (SURVEY §8 header).
SEQLEN = 16          # T
FEAT_DIM = 2048      # F, per-frame image feature
NUM_VERTS = 431      # V, twice-downsampled SMPL mesh
NUM_VERTS_FULL = 6890
GRU_HIDDEN = 1024
LIFTER_HEADS = 8
JOINT_HEADS = 8      # CoevoDecoder.py:139
VERTX_HEADS = 2      # CoevoDecoder.py:140
NUM_TEMPLATE_JOINTS = 17   # H36M-regressed template joints used by vj_relation (CoevoDecoder.py:207-209)
