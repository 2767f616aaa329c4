"""Checkpoint -> packed device tensors.

Takes a reference-layout ``state_dict`` (keys of lib/models/PMCE.py:11-13) — optionally wrapped
in the reference checkpoint dict (``{'model_state_dict': ...}``, lib/core/base.py:66-67) and optionally with the
``module.`` prefix of nn.DataParallel (lib/funcs_utils.py:65-70) — and lays the weights out the way the HIP
kernels read them.  Pure layout work (concatenate / transpose / zero-pad), done once at load time with torch
ops on whatever device the tensors live on; no arithmetic of the forward happens here except the three
init-time embedding sums (Eq, Ev, Ek) which the reference recomputes every forward.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch

from .config import FEAT_DIM, GRU_HIDDEN, NUM_VERTS, NUM_VERTS_FULL, SEQLEN

FINAL_K = 3360            # 2048 + 3*431 = 3341, rounded up to a multiple of 32 (GEMM k-tile)
N_ADA = 24                # live AdaLN instances

# order of the AdaLN instances in the packed gamma/beta product (must match csrc/model.cpp)
ADA_ORDER = (
    [f"coevoblock{k}.{m}" for k in (1, 2, 3) for m in
     ("vertx_CA_FFN.normq", "vertx_CA_FFN.normk", "vertx_CA_FFN.normv", "vertx_CA_FFN.norm2",
      "vertx_SA_FFN.norm1", "vertx_SA_FFN.norm2")]
    + [f"coevoblock3.{m}" for m in
       ("joint_CA_FFN.normq", "joint_CA_FFN.normk", "joint_CA_FFN.normv", "joint_CA_FFN.norm2",
        "joint_SA_FFN.norm1", "joint_SA_FFN.norm2")]
)
assert len(ADA_ORDER) == N_ADA


def unwrap_checkpoint(obj):
    """Accept the reference's checkpoint dict or a bare state_dict; strip a DataParallel 'module.' prefix."""
    sd = obj["model_state_dict"] if isinstance(obj, dict) and "model_state_dict" in obj else obj
    if any(k.startswith("module.") for k in sd.keys()):
        sd = OrderedDict((k[len("module."):] if k.startswith("module.") else k, v) for k, v in sd.items())
    return sd


def _f32(t, device):
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


def pack_lifter(sd, prefix, device, num_joint, embed_dim, depth):
    """GraphormerNet weights are used as stored (PoseEstimation.py:31-66); only renamed and squeezed."""
    out = OrderedDict()
    C = embed_dim

    def g(k):
        return _f32(sd[prefix + k], device)

    out["lifter.joint_embed.weight"] = g("joint_embed.weight")
    out["lifter.joint_embed.bias"] = g("joint_embed.bias")
    out["lifter.imgfeat_embed.weight"] = g("imgfeat_embed.weight")
    out["lifter.imgfeat_embed.bias"] = g("imgfeat_embed.bias")
    out["lifter.spatial_pos_embed"] = g("spatial_pos_embed").reshape(num_joint, C).contiguous()
    out["lifter.temporal_pos_embed"] = g("temporal_pos_embed").reshape(SEQLEN, C).contiguous()
    for kind in ("Spatial", "Temporal"):
        for i in range(depth):
            for leaf in ("norm1.weight", "norm1.bias", "attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight",
                         "attn.proj.bias", "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias",
                         "mlp.fc2.weight", "mlp.fc2.bias"):
                out[f"lifter.{kind}Blocks.{i}.{leaf}"] = g(f"{kind}Blocks.{i}.{leaf}")
    for k in ("norm_s.weight", "norm_s.bias", "norm_t.weight", "norm_t.bias", "regression.0.weight",
              "regression.0.bias", "regression.1.weight", "regression.1.bias"):
        out["lifter." + k] = g(k)
    out["lifter.fusion.weight"] = g("fusion.weight").reshape(SEQLEN).contiguous()
    out["lifter.fusion.bias"] = g("fusion.bias").reshape(1).contiguous()
    return out


def pack_final(sd, prefix, device):
    """Conv1d(431->6890,k=3,pad=1) along xyz + the three Linear(2048->6890) as ONE [20670, 3360] operand.
    Output column n = 3*o + l (so the product IS the [B,6890,3] mesh); K = [relu(g) (2048) | vt[c][l'] (1293) | 0].
    out[b,o,l] = sum_f relu(g)[b,f] Wl[o,f] + sum_{c,l'} vt[b,c,l'] Wc[o,c,l'-l+1]  (CoevoDecoder.py:238-244)."""
    Wc = _f32(sd[prefix + "upsample_conv.weight"], device)            # [6890, 431, 3]
    bc = _f32(sd[prefix + "upsample_conv.bias"], device)
    W = torch.zeros(NUM_VERTS_FULL, 3, FINAL_K, device=device, dtype=torch.float32)
    b = torch.zeros(NUM_VERTS_FULL, 3, device=device, dtype=torch.float32)
    for l in range(3):
        W[:, l, :FEAT_DIM] = _f32(sd[f"{prefix}linear_cur{l + 1}.weight"], device)
        b[:, l] = _f32(sd[f"{prefix}linear_cur{l + 1}.bias"], device) + bc
        conv = torch.zeros(NUM_VERTS_FULL, NUM_VERTS, 3, device=device, dtype=torch.float32)
        for lp in range(3):
            k = lp - l + 1
            if 0 <= k <= 2:
                conv[:, :, lp] = Wc[:, :, k]
        W[:, l, FEAT_DIM:FEAT_DIM + NUM_VERTS * 3] = conv.reshape(NUM_VERTS_FULL, NUM_VERTS * 3)
    return W.reshape(NUM_VERTS_FULL * 3, FINAL_K).contiguous(), b.reshape(NUM_VERTS_FULL * 3).contiguous()


def pack_decoder(sd, prefix, device, num_joint, vj_relation):
    out = OrderedDict()
    J = num_joint

    def g(k):
        return _f32(sd[prefix + k], device)

    vj = np.asarray(vj_relation).astype(np.int64)
    assert vj.shape == (NUM_VERTS,) and vj.min() >= 0 and vj.max() < 17 <= J
    out["dec.vj_relation"] = torch.from_numpy(vj.astype(np.int32)).to(device)
    # ---- GRU (nn.GRU parameter names, CoevoDecoder.py:216-221): concatenate / stack the directions
    out["dec.gru.w_ih_l0"] = torch.cat([g("gru_cur.weight_ih_l0"), g("gru_cur.weight_ih_l0_reverse")], 0).contiguous()
    out["dec.gru.b_ih_l0"] = torch.cat([g("gru_cur.bias_ih_l0"), g("gru_cur.bias_ih_l0_reverse")], 0).contiguous()
    for layer in (0, 1):
        out[f"dec.gru.w_hh_l{layer}"] = torch.stack(
            [g(f"gru_cur.weight_hh_l{layer}"), g(f"gru_cur.weight_hh_l{layer}_reverse")], 0).contiguous()
        out[f"dec.gru.b_hh_l{layer}"] = torch.stack(
            [g(f"gru_cur.bias_hh_l{layer}"), g(f"gru_cur.bias_hh_l{layer}_reverse")], 0).contiguous()
    out["dec.gru.w_ih_l1"] = torch.stack([g("gru_cur.weight_ih_l1"), g("gru_cur.weight_ih_l1_reverse")], 0).contiguous()
    out["dec.gru.b_ih_l1"] = torch.stack([g("gru_cur.bias_ih_l1"), g("gru_cur.bias_ih_l1_reverse")], 0).contiguous()
    # ---- AdaLN gamma/beta Linear(2048->64) layers, live instances only
    ws, bs = [], []
    for name in ADA_ORDER:
        ws += [g(name + ".mlp_gamma.weight"), g(name + ".mlp_beta.weight")]
        bs += [g(name + ".mlp_gamma.bias"), g(name + ".mlp_beta.bias")]
    out["dec.ada.weight"] = torch.cat(ws, 0).contiguous()          # [3072, 2048]
    out["dec.ada.bias"] = torch.cat(bs, 0).contiguous()
    # ---- co-evolution blocks
    for k in (1, 2, 3):
        p, q = f"coevoblock{k}.", f"dec.b{k}."
        out[q + "joint_proj.weight"] = g(p + "joint_proj.weight")
        out[q + "joint_proj.bias"] = g(p + "joint_proj.bias")
        out[q + "joint_pos_embed"] = g(p + "joint_pos_embed").reshape(J, 64).contiguous()
        out[q + "proj_j2v_dim.weight"] = g(p + "proj_j2v_dim.weight")
        out[q + "proj_j2v_dim.bias"] = g(p + "proj_j2v_dim.bias")
        out[q + "j2v_K_embed"] = g(p + "j2v_K_embed").reshape(J, 64).contiguous()
        out[q + "vertx_proj.weight"] = g(p + "vertx_proj.weight")
        vpos = g(p + "vertx_pos_embed").reshape(NUM_VERTS, 64)
        ev = g(p + "vertx_proj.bias")[None, :] + vpos                # (proj bias + pos), CoevoDecoder.py:177-180
        out[q + "Eq"] = (ev + g(p + "v_Q_embed").reshape(NUM_VERTS, 64)).contiguous()   # + v_Q_embed (:184)
        for w in ("wq", "wk", "wv", "proj"):
            out[f"{q}vca.{w}.weight"] = g(f"{p}vertx_CA_FFN.attn.{w}.weight")
            out[f"{q}vca.{w}.bias"] = g(f"{p}vertx_CA_FFN.attn.{w}.bias")
        for w in ("fc1", "fc2"):
            out[f"{q}vca.mlp.{w}.weight"] = g(f"{p}vertx_CA_FFN.mlp.{w}.weight")
            out[f"{q}vca.mlp.{w}.bias"] = g(f"{p}vertx_CA_FFN.mlp.{w}.bias")
        for w in ("qkv", "proj"):
            out[f"{q}vsa.{w}.weight"] = g(f"{p}vertx_SA_FFN.attn.{w}.weight")
            out[f"{q}vsa.{w}.bias"] = g(f"{p}vertx_SA_FFN.attn.{w}.bias")
        for w in ("fc1", "fc2"):
            out[f"{q}vsa.mlp.{w}.weight"] = g(f"{p}vertx_SA_FFN.mlp.{w}.weight")
            out[f"{q}vsa.mlp.{w}.bias"] = g(f"{p}vertx_SA_FFN.mlp.{w}.bias")
        out[q + "vcoor.weight"] = g(p + "proj_vertx_feat2coor.weight")
        out[q + "vcoor.bias"] = g(p + "proj_vertx_feat2coor.bias")
        if k == 3:   # joint stream: live only here
            out[q + "Ev"] = ev.contiguous()
            out[q + "proj_v2j_dim.weight"] = g(p + "proj_v2j_dim.weight")
            out[q + "Ek"] = (g(p + "proj_v2j_dim.bias")[None, :] + g(p + "v2j_K_embed").reshape(NUM_VERTS, 64)).contiguous()
            out[q + "j_Q_embed"] = g(p + "j_Q_embed").reshape(J, 64).contiguous()
            for w in ("wq", "wk", "wv", "proj"):
                out[f"{q}jca.{w}.weight"] = g(f"{p}joint_CA_FFN.attn.{w}.weight")
                out[f"{q}jca.{w}.bias"] = g(f"{p}joint_CA_FFN.attn.{w}.bias")
            for w in ("fc1", "fc2"):
                out[f"{q}jca.mlp.{w}.weight"] = g(f"{p}joint_CA_FFN.mlp.{w}.weight")
                out[f"{q}jca.mlp.{w}.bias"] = g(f"{p}joint_CA_FFN.mlp.{w}.bias")
            for w in ("qkv", "proj"):
                out[f"{q}jsa.{w}.weight"] = g(f"{p}joint_SA_FFN.attn.{w}.weight")
                out[f"{q}jsa.{w}.bias"] = g(f"{p}joint_SA_FFN.attn.{w}.bias")
            for w in ("fc1", "fc2"):
                out[f"{q}jsa.mlp.{w}.weight"] = g(f"{p}joint_SA_FFN.mlp.{w}.weight")
                out[f"{q}jsa.mlp.{w}.bias"] = g(f"{p}joint_SA_FFN.mlp.{w}.bias")
            out[q + "jcoor.weight"] = g(p + "proj_joint_feat2coor.weight")
            out[q + "jcoor.bias"] = g(p + "proj_joint_feat2coor.bias")
    out["dec.final.weight"], out["dec.final.bias"] = pack_final(sd, prefix, device)
    return out


def pack_regressor(j_regressor, device):
    """Dense [R,6890] regressor -> CSR device tensors for pmce_j_regress (lib/core/base.py:196,225)."""
    from .assets import regressor_to_csr
    indptr, indices, data = regressor_to_csr(np.asarray(j_regressor))
    return OrderedDict([
        ("jreg.indptr", torch.from_numpy(indptr).to(device)),
        ("jreg.indices", torch.from_numpy(indices).to(device)),
        ("jreg.data", torch.from_numpy(data).to(device)),
    ]), int(np.asarray(j_regressor).shape[0])


def expected_shapes(num_joint, embed_dim, depth):
    """Shapes the state_dict must have (validation at load)."""
    from .synth import pmce_spec
    return OrderedDict((k, tuple(v[0])) for k, v in pmce_spec(num_joint, embed_dim, depth).items())
