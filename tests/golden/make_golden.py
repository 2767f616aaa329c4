#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference (imported from /root/reference/lib, never
copied) on the deterministic synthetic weights/inputs of pmce_amd.synth.

Runs only in the build container (the GPU box has no /root/reference).  Usage:
    python tests/golden/make_golden.py            # writes tests/golden/*.npz

Import shims — everything below is this repo's own code:
  * timm is not installed: stub ``timm.models.layers.{DropPath,to_2tuple,trunc_normal_}`` and
    ``timm.models.vision_transformer.{_cfg,Mlp,Attention}`` with the published timm semantics
    (Attention = the in-tree copy at CoevoDecoder.py:107-131; Mlp = fc1 -> GELU -> fc2).
  * easydict not installed; ``core.config`` has import side effects -> replaced by a stub module.
  * ``funcs_utils`` imports cv2 -> stub with load_checkpoint only.
  * no GPU: ``Tensor.cuda`` -> identity, ``device('cuda')`` -> cpu.
  * data/base_data is a dangling symlink -> a temp cwd with synthetic stand-ins (pmce_amd.synth.make_base_data)
    and the reference's own J_regressor_h36m_correct.npy.
Fixtures hold inputs' seeds/checksums and the reference's OUTPUTS only (data, no code).
"""
import os
import os.path as osp
import shutil
import sys
import tempfile
import types

import numpy as np
import torch
import torch.nn as nn

HERE = osp.dirname(osp.abspath(__file__))
REPO = osp.dirname(osp.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)

from pmce_amd import synth  # noqa: E402


def install_shims():
    # ---- timm -------------------------------------------------------------------------------
    timm = types.ModuleType("timm")
    tm = types.ModuleType("timm.models")
    tl = types.ModuleType("timm.models.layers")
    tv = types.ModuleType("timm.models.vision_transformer")

    class DropPath(nn.Module):
        def __init__(self, drop_prob=0.0):
            super().__init__()
            self.drop_prob = drop_prob

        def forward(self, x):
            assert not self.training, "golden generation is eval()-only"
            return x

    class Mlp(nn.Module):
        def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
            super().__init__()
            out_features = out_features or in_features
            hidden_features = hidden_features or in_features
            self.fc1 = nn.Linear(in_features, hidden_features)
            self.act = act_layer()
            self.fc2 = nn.Linear(hidden_features, out_features)
            self.drop = nn.Dropout(drop)

        def forward(self, x):
            return self.drop(self.fc2(self.drop(self.act(self.fc1(x)))))

    class Attention(nn.Module):
        def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0.0, proj_drop=0.0):
            super().__init__()
            self.num_heads = num_heads
            self.scale = (dim // num_heads) ** -0.5
            self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
            self.attn_drop = nn.Dropout(attn_drop)
            self.proj = nn.Linear(dim, dim)
            self.proj_drop = nn.Dropout(proj_drop)

        def forward(self, x):
            B, N, C = x.shape
            qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
            q, k, v = qkv.unbind(0)
            attn = (q @ k.transpose(-2, -1)) * self.scale
            attn = self.attn_drop(attn.softmax(dim=-1))
            x = (attn @ v).transpose(1, 2).reshape(B, N, C)
            return self.proj_drop(self.proj(x))

    tl.DropPath = DropPath
    tl.to_2tuple = lambda x: (x, x)
    tl.trunc_normal_ = nn.init.trunc_normal_
    tv._cfg = lambda **kw: kw
    tv.Mlp = Mlp
    tv.Attention = Attention
    timm.models = tm
    tm.layers = tl
    tm.vision_transformer = tv
    sys.modules.update({"timm": timm, "timm.models": tm, "timm.models.layers": tl,
                        "timm.models.vision_transformer": tv})

    # ---- core.config / funcs_utils ------------------------------------------------------------
    class AD(dict):
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__

    core = types.ModuleType("core")
    cc = types.ModuleType("core.config")
    cfg = AD()
    cfg.DATASET = AD(seqlen=16, BASE_DATA_DIR="data/base_data")
    cfg.MODEL = AD(joint_dim=64, vertx_dim=64, posenet_pretrained=False, posenet_path="", hpe_dim=256, hpe_dep=3)
    cc.cfg = cfg
    core.config = cc
    sys.modules.update({"core": core, "core.config": cc})
    fu = types.ModuleType("funcs_utils")
    fu.load_checkpoint = lambda load_dir, epoch=0, pick_best=False: torch.load(load_dir, map_location="cpu")
    sys.modules["funcs_utils"] = fu

    # ---- no GPU -------------------------------------------------------------------------------
    torch.Tensor.cuda = lambda self, *a, **k: self
    _orig_to = torch.Tensor.to

    def _to(self, *a, **k):
        a = tuple(torch.device("cpu") if isinstance(x, torch.device) and x.type == "cuda" else x for x in a)
        if isinstance(k.get("device"), torch.device) and k["device"].type == "cuda":
            k["device"] = torch.device("cpu")
        return _orig_to(self, *a, **k)

    torch.Tensor.to = _to
    sys.path.insert(0, osp.join(REF, "lib"))


def make_cwd():
    import scipy.sparse as sp
    d = tempfile.mkdtemp(prefix="pmce_golden_")
    os.makedirs(osp.join(d, "data", "base_data"))
    os.makedirs(osp.join(d, "data", "Human36M"))
    v, D = synth.make_base_data()
    np.save(osp.join(d, "data", "base_data", "smpl_mean_vertices.npy"), v)
    tiny = sp.identity(4, format="csr", dtype=np.float32)
    A = np.empty(3, dtype=object)
    U = np.empty(2, dtype=object)
    Dd = np.empty(2, dtype=object)
    for i in range(3):
        A[i] = tiny
    for i in range(2):
        U[i] = tiny
        Dd[i] = D[i]
    np.savez(osp.join(d, "data", "base_data", "mesh_downsampling.npz"), A=A, U=U, D=Dd)
    shutil.copy(osp.join(REF, "data", "Human36M", "J_regressor_h36m_correct.npy"),
                osp.join(d, "data", "Human36M", "J_regressor_h36m_correct.npy"))
    return d


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def n(x):
    return x.detach().cpu().numpy().astype(np.float32)


def main():
    torch.set_num_threads(8)
    install_shims()
    cwd = make_cwd()
    os.chdir(cwd)
    import models  # noqa: F401  (reference lib/models/__init__.py)
    from models import CoevoDecoder as RC

    out_dir = HERE
    meta = {}
    with torch.no_grad():
        # ---------------- end-to-end -----------------------------------------------------------
        for (J, C, B, seed) in [(17, 256, 2, 0), (19, 256, 1, 3), (17, 512, 1, 5)]:
            model = models.PMCE.get_model(J, C, 3).eval()
            template431 = n(model.pose_mesh_coevo.init_vertices)     # before the checkpoint overwrites the buffer
            sd = synth.make_state_dict(synth.pmce_spec(J, C, 3), seed=123)
            ref_keys = set(model.state_dict().keys())
            assert ref_keys == set(sd.keys()), (sorted(ref_keys ^ set(sd.keys()))[:10])
            for k, v in model.state_dict().items():
                assert tuple(v.shape) == tuple(sd[k].shape), (k, v.shape, sd[k].shape)
            model.load_state_dict(sd, strict=True)
            pose2d, img_feat = synth.make_inputs(B, J, seed)
            cam_mesh, cam_pose, pose3d = model(t(pose2d), t(img_feat))
            jreg = np.load("data/Human36M/J_regressor_h36m_correct.npy")
            J_regressor = torch.Tensor(jreg)                                   # base.py:196
            pred_pose = torch.matmul(J_regressor[None, :, :], cam_mesh * 1000)  # base.py:223-225
            dec = model.pose_mesh_coevo
            vj = np.asarray(dec.vj_relation).astype(np.int64)
            # intermediates through the reference's own submodules
            y, _ = dec.gru_cur(t(img_feat).permute(1, 0, 2))
            g = y[8]
            joints = pose3d / 1000
            vert0 = joints[:, dec.vj_relation, :3]
            _, v1 = dec.coevoblock1(joints, vert0, g)
            _, v2 = dec.coevoblock2(joints, v1, g)
            j3, v3 = dec.coevoblock3(joints, v2, g)
            assert torch.equal(j3, cam_pose)
            np.savez_compressed(
                osp.join(out_dir, f"e2e_J{J}_C{C}_B{B}.npz"),
                J=J, C=C, B=B, input_seed=seed, weight_seed=123,
                pose2d_sum=float(pose2d.astype(np.float64).sum()), img_feat_sum=float(img_feat.astype(np.float64).sum()),
                cam_mesh=n(cam_mesh), cam_pose=n(cam_pose), pose3d=n(pose3d), pred_pose=n(pred_pose),
                vj_relation=vj, init_vertices=template431, g_mid=n(g), vert0=n(vert0),
                v1=n(v1), v2=n(v2), v3=n(v3))
            print(f"e2e J={J} C={C} B={B}: |pose3d|max={pose3d.abs().max():.1f}mm |mesh|max={cam_mesh.abs().max():.3f}m")
            if (J, C) == (17, 256):
                ref17 = model
                sd17 = sd

        # ---------------- per-module (decoder, J=17) ----------------------------------------------
        dec = ref17.pose_mesh_coevo
        blk = dec.coevoblock3
        B = 1
        u = synth.uniform_pm1
        g = t(u("mod.g", B * 2048, 11).reshape(B, 2048) * 0.8)
        xv = t(u("mod.xv", B * 431 * 64, 11).reshape(B, 431, 64) * 1.5 + 0.1)
        xj = t(u("mod.xj", B * 17 * 64, 11).reshape(B, 17, 64) * 1.5 - 0.2)
        mods = {}
        mods["adaln_v"] = blk.vertx_CA_FFN.normq(xv, g)
        ca = blk.vertx_CA_FFN
        mods["ca_v_from_j"] = xv + ca.attn(ca.normq(xv, g), ca.normk(xj, g), ca.normv(xj, g))   # CoevoDecoder.py:83
        mods["cab_v_from_j"] = ca(xv, xj, xj, g)
        cj = blk.joint_CA_FFN
        mods["ca_j_from_v"] = xj + cj.attn(cj.normq(xj, g), cj.normk(xv, g), cj.normv(xv, g))
        mods["cab_j_from_v"] = cj(xj, xv, xv, g)
        mods["sab_v"] = blk.vertx_SA_FFN(xv, g)
        mods["sab_j"] = blk.joint_SA_FFN(xj, g)
        jt = t(u("mod.jt", B * 17 * 3, 11).reshape(B, 17, 3) * 0.5)
        vt = t(u("mod.vt", B * 431 * 3, 11).reshape(B, 431, 3) * 0.5)
        jo, vo = blk(jt, vt, g)
        mods["coevo_j"], mods["coevo_v"] = jo, vo
        mods["upsample"] = dec.upsample_conv(vt)
        feats = t(synth.make_inputs(2, 17, 21)[1])
        y, _ = dec.gru_cur(feats.permute(1, 0, 2))
        mods["gru_y8"] = y[8]
        mods["gru_y_all_b0"] = y[:, 0, :]
        # lifter alone (LiftTester path, base.py:357) + one lifter block
        p2d, f2 = synth.make_inputs(1, 17, 31)
        mods["lifter_pose3d"] = ref17.pose_lifter(t(p2d), t(f2))
        xl = t(u("mod.xl", 3 * 17 * 256, 11).reshape(3, 17, 256))
        mods["lifter_block_s1"] = ref17.pose_lifter.SpatialBlocks[1](xl)
        np.savez_compressed(osp.join(out_dir, "modules_J17_C256.npz"), seed=11,
                            **{k: n(v) for k, v in mods.items()})
        for k, v in mods.items():
            print(f"  module {k}: {tuple(v.shape)} |max|={v.abs().max():.4f}")
    os.chdir(REPO)
    shutil.rmtree(cwd, ignore_errors=True)
    print("golden fixtures written to", out_dir)


if __name__ == "__main__":
    main()
