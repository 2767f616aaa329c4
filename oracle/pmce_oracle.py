"""ORACLE — TEST INFRASTRUCTURE ONLY.  A CPU restatement (plain torch ops on CPU tensors, fp32 by
default, fp64 on request) of the reference's per-clip inference forward.  Nothing under pmce_amd/
imports this file; only tests/, __graft_entry__.smoke() and bench.py's ``cpu_baseline`` leg do, and
only as the checker / the reported CPU baseline.

Parity pin: the reference has no tests or golden vectors of its own, so this oracle is
pinned against outputs of the reference itself — tests/golden/*.npz, generated in the build
container by tests/golden/make_golden.py, which imports the real /root/reference/lib/models with
import shims and runs its CPU forward on the same deterministic weights/inputs.
tests/test_oracle_golden.py checks every function here against those fixtures.

Third-party arithmetic: ``timm.models.vision_transformer.{Attention,Mlp}`` and
``timm.models.layers.DropPath`` are imported by the reference (PoseEstimation.py:9-10,
CoevoDecoder.py:6-7) but timm is neither vendored nor version-pinned (requirements.sh has no timm
line).  Their published algorithm is restated in ``self_attention``/``mlp`` below; the in-tree copy
of the same Attention (CoevoDecoder.py:107-131) is the authority for the formula.  DropPath/Dropout
are identity in eval().

All functions take the reference-layout ``state_dict`` and a key prefix.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

T_SEQ = 16


def _g(sd, key, dtype):
    return sd[key].to(dtype)


def linear(x, sd, p, dtype):
    return F.linear(x, _g(sd, p + ".weight", dtype), _g(sd, p + ".bias", dtype))


def layer_norm(x, sd, p, eps, dtype):
    return F.layer_norm(x, (x.shape[-1],), _g(sd, p + ".weight", dtype), _g(sd, p + ".bias", dtype), eps)


def self_attention(x, sd, p, num_heads, dtype):
    """timm Attention == CoevoDecoder.py:118-131: qkv Linear -> [B,N,3,H,hd] -> softmax(q k^T hd^-.5) v -> proj."""
    B, N, C = x.shape
    hd = C // num_heads
    qkv = linear(x, sd, p + ".qkv", dtype).reshape(B, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q @ k.transpose(-2, -1)) * (hd ** -0.5)
    attn = attn.softmax(dim=-1)
    out = (attn @ v).transpose(1, 2).reshape(B, N, C)
    return linear(out, sd, p + ".proj", dtype)


def mlp(x, sd, p, dtype):
    """timm Mlp: fc1 -> exact (erf) GELU -> fc2."""
    return linear(F.gelu(linear(x, sd, p + ".fc1", dtype)), sd, p + ".fc2", dtype)


# ------------------------------------------------------------------------------------------------
# temporal pose encoder ("lifter") — reference lib/models/PoseEstimation.py
# ------------------------------------------------------------------------------------------------

def lifter_block(x, sd, p, dtype):
    """PoseEstimation.py:13-29: pre-LN block, LN eps 1e-6 (:38), 8 heads, mlp ratio 2."""
    x = x + self_attention(layer_norm(x, sd, p + ".norm1", 1e-6, dtype), sd, p + ".attn", 8, dtype)
    x = x + mlp(layer_norm(x, sd, p + ".norm2", 1e-6, dtype), sd, p + ".mlp", dtype)
    return x


def lifter_forward(sd, pose2d, img_feat, depth=3, prefix="pose_lifter.", dtype=torch.float32,
                   return_tokens=False):
    """GraphormerNet.forward (PoseEstimation.py:95-115) incl. SpaTemHead (:76-93).
    pose2d[B,T,J,2], img_feat[B,T,2048] -> pose3d[B,J,3] (mm)."""
    pose2d = pose2d.to(dtype)
    img_feat = img_feat.to(dtype)
    b, t, j, _ = pose2d.shape
    x = pose2d.reshape(b * t, j, 2)                                          # :78
    x = linear(x, sd, prefix + "joint_embed", dtype)                         # :79
    x = x + linear(img_feat, sd, prefix + "imgfeat_embed", dtype).reshape(b * t, 1, -1)   # :80
    x = x + _g(sd, prefix + "spatial_pos_embed", dtype)                      # :81
    c = x.shape[-1]
    for i in range(depth):
        # spatial: sequences = (b t), tokens = j                               :83-85 / :101-103
        x = lifter_block(x, sd, f"{prefix}SpatialBlocks.{i}", dtype)
        x = layer_norm(x, sd, prefix + "norm_s", 1e-6, dtype)
        # '(b t) j c -> (b j) t c'                                              :87 / :104
        x = x.reshape(b, t, j, c).permute(0, 2, 1, 3).reshape(b * j, t, c)
        if i == 0:
            x = x + _g(sd, prefix + "temporal_pos_embed", dtype)             # :88 (first time only)
        x = lifter_block(x, sd, f"{prefix}TemporalBlocks.{i}", dtype)        # :90-91 / :105
        x = layer_norm(x, sd, prefix + "norm_t", 1e-6, dtype)                # :92 / :106
        if i + 1 < depth:
            # '(b j) t c -> (b t) j c'                                          :101
            x = x.reshape(b, j, t, c).permute(0, 2, 1, 3).reshape(b * t, j, c)
    x = x.reshape(b, j, t, c).permute(0, 2, 1, 3)                            # :109  b t j c
    tokens = x
    x = layer_norm(x, sd, prefix + "regression.0", 1e-5, dtype)              # :62-65 nn.LayerNorm default eps
    x = linear(x, sd, prefix + "regression.1", dtype)                        # [b,t,j,3]
    w = _g(sd, prefix + "fusion.weight", dtype).reshape(1, t, 1, 1)          # Conv2d(T->1,k=1)  :66,112
    out = (x * w).sum(1) + _g(sd, prefix + "fusion.bias", dtype)             # [b,j,3]
    if return_tokens:
        return out, tokens
    return out


# ------------------------------------------------------------------------------------------------
# pose–mesh co-evolution decoder — reference lib/models/CoevoDecoder.py
# ------------------------------------------------------------------------------------------------

def ada_layer_norm(x, img_feat, sd, p, dtype, eps=1e-6):
    """AdaLayerNorm.forward (CoevoDecoder.py:23-29): unbiased std, eps added to the std."""
    mean = x.mean(-1, keepdim=True)
    std = x.std(-1, keepdim=True)                                 # Bessel-corrected
    gamma = linear(img_feat, sd, p + ".mlp_gamma", dtype).unsqueeze(1)
    beta = linear(img_feat, sd, p + ".mlp_beta", dtype).unsqueeze(1)
    return gamma * (x - mean) / (std + eps) + beta


def cross_attention(xq, xk, xv, sd, p, num_heads, dtype):
    """CrossAttention.forward (CoevoDecoder.py:47-62)."""
    B, N, C = xq.shape
    M = xk.shape[1]
    hd = C // num_heads
    q = linear(xq, sd, p + ".wq", dtype).reshape(B, N, num_heads, hd).permute(0, 2, 1, 3)
    k = linear(xk, sd, p + ".wk", dtype).reshape(B, M, num_heads, hd).permute(0, 2, 1, 3)
    v = linear(xv, sd, p + ".wv", dtype).reshape(B, M, num_heads, hd).permute(0, 2, 1, 3)
    attn = (q @ k.transpose(-2, -1)) * (hd ** -0.5)
    attn = attn.softmax(dim=-1)
    x = (attn @ v).transpose(1, 2).reshape(B, N, C)
    return linear(x, sd, p + ".proj", dtype)


def cross_attention_only(xq, xk, xv, g, sd, p, num_heads, dtype=torch.float32):
    """First half of CrossAttentionBlock.forward (CoevoDecoder.py:83): the fused AdaLN + cross-attention
    + residual unit that the north-star HIP kernel implements."""
    return xq + cross_attention(ada_layer_norm(xq, g, sd, p + ".normq", dtype),
                                ada_layer_norm(xk, g, sd, p + ".normk", dtype),
                                ada_layer_norm(xv, g, sd, p + ".normv", dtype), sd, p + ".attn", num_heads, dtype)


def cross_attention_block(xq, xk, xv, g, sd, p, num_heads, dtype=torch.float32):
    """CrossAttentionBlock.forward (CoevoDecoder.py:82-87)."""
    xq = cross_attention_only(xq, xk, xv, g, sd, p, num_heads, dtype)
    xq = xq + mlp(ada_layer_norm(xq, g, sd, p + ".norm2", dtype), sd, p + ".mlp", dtype)
    return xq


def ada_block(x, g, sd, p, num_heads, dtype=torch.float32):
    """AdaLN Block.forward (CoevoDecoder.py:102-105)."""
    x = x + self_attention(ada_layer_norm(x, g, sd, p + ".norm1", dtype), sd, p + ".attn", num_heads, dtype)
    x = x + mlp(ada_layer_norm(x, g, sd, p + ".norm2", dtype), sd, p + ".mlp", dtype)
    return x


def coevo_block(joint, vertx, g, sd, p, dtype=torch.float32):
    """CoevoBlock.forward (CoevoDecoder.py:175-191).  Both cross-attention updates read the PRE-update
    features (tuple right-hand side, :183-184)."""
    jf = linear(joint, sd, p + ".joint_proj", dtype) + _g(sd, p + ".joint_pos_embed", dtype)    # :177-180
    vf = linear(vertx, sd, p + ".vertx_proj", dtype) + _g(sd, p + ".vertx_pos_embed", dtype)
    jf_new = cross_attention_block(jf + _g(sd, p + ".j_Q_embed", dtype),
                                   linear(vf, sd, p + ".proj_v2j_dim", dtype) + _g(sd, p + ".v2j_K_embed", dtype),
                                   vf, g, sd, p + ".joint_CA_FFN", 8, dtype)                   # :183
    vf_new = cross_attention_block(vf + _g(sd, p + ".v_Q_embed", dtype),
                                   linear(jf, sd, p + ".proj_j2v_dim", dtype) + _g(sd, p + ".j2v_K_embed", dtype),
                                   jf, g, sd, p + ".vertx_CA_FFN", 2, dtype)                   # :184
    jf = ada_block(jf_new, g, sd, p + ".joint_SA_FFN", 8, dtype)                               # :187
    vf = ada_block(vf_new, g, sd, p + ".vertx_SA_FFN", 2, dtype)
    joint_out = linear(jf, sd, p + ".proj_joint_feat2coor", dtype) + joint[:, :, :3]           # :189
    vertx_out = linear(vf, sd, p + ".proj_vertx_feat2coor", dtype) + vertx[:, :, :3]
    return joint_out, vertx_out


def gru_bidir2(x_seq, sd, p, dtype=torch.float32):
    """nn.GRU(2048,1024,bidirectional=True,num_layers=2), seq-first, h0 = 0 (CoevoDecoder.py:216-221,228).
    PyTorch gate order (r,z,n); n = tanh(W_in x + b_in + r*(W_hn h + b_hn)); h' = (1-z)*n + z*h.
    x_seq[T,B,2048] -> y[T,B,2048] (top layer, fwd|bwd)."""
    Tn, B, _ = x_seq.shape
    inp = x_seq
    H = 1024
    for layer in (0, 1):
        outs = []
        for sfx in ("", "_reverse"):
            w_ih = _g(sd, f"{p}.weight_ih_l{layer}{sfx}", dtype)
            w_hh = _g(sd, f"{p}.weight_hh_l{layer}{sfx}", dtype)
            b_ih = _g(sd, f"{p}.bias_ih_l{layer}{sfx}", dtype)
            b_hh = _g(sd, f"{p}.bias_hh_l{layer}{sfx}", dtype)
            gi_all = F.linear(inp, w_ih, b_ih)                     # [T,B,3H]
            h = torch.zeros(B, H, dtype=dtype)
            ys = [None] * Tn
            order = range(Tn) if sfx == "" else range(Tn - 1, -1, -1)
            for t in order:
                gi = gi_all[t]
                gh = F.linear(h, w_hh, b_hh)
                r = torch.sigmoid(gi[:, :H] + gh[:, :H])
                z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
                n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
                h = (1 - z) * n + z * h
                ys[t] = h
            outs.append(torch.stack(ys, 0))
        inp = torch.cat(outs, -1)
    return inp


def vertex_init_gather(joints, vj_relation):
    """vertxs = joints[:, vj_relation, :3] (CoevoDecoder.py:232) — integer index gather, bit-exact."""
    idx = torch.as_tensor(np.asarray(vj_relation).astype(np.int64))
    return joints[:, idx, :3]


def upsample_and_residual(vertxs, g_mid, sd, prefix, dtype=torch.float32):
    """CoevoDecoder.py:238-244: Conv1d(431->6890,k=3,pad=1) along xyz + cat of 3 Linear(relu(y[8]))."""
    vertxs, g_mid = vertxs.to(dtype), g_mid.to(dtype)
    up = F.conv1d(vertxs, _g(sd, prefix + "upsample_conv.weight", dtype), _g(sd, prefix + "upsample_conv.bias", dtype),
                  padding=1)                                       # [B,6890,3]
    rg = F.relu(g_mid)
    res = torch.stack([linear(rg, sd, f"{prefix}linear_cur{i}", dtype) for i in (1, 2, 3)], -1)
    return up + res


def decoder_forward(sd, joints, img_feats, vj_relation, prefix="pose_mesh_coevo.", dtype=torch.float32,
                    return_intermediates=False):
    """Pose2Mesh.forward (CoevoDecoder.py:226-246). joints[B,J,3] (m), img_feats[B,16,2048]
    -> (joints3[B,J,3], mesh[B,6890,3])."""
    joints = joints.to(dtype)
    img_feats = img_feats.to(dtype)
    y = gru_bidir2(img_feats.permute(1, 0, 2), sd, prefix + "gru_cur", dtype)   # :228
    g = y[T_SEQ // 2]                                                            # :229
    vert0 = vertex_init_gather(joints, vj_relation)                              # :232
    j1, v1 = coevo_block(joints, vert0, g, sd, prefix + "coevoblock1", dtype)    # :235
    j2, v2 = coevo_block(joints, v1, g, sd, prefix + "coevoblock2", dtype)       # :236  (original joints!)
    j3, v3 = coevo_block(joints, v2, g, sd, prefix + "coevoblock3", dtype)       # :237
    mesh = upsample_and_residual(v3, g, sd, prefix, dtype)                       # :238-244
    if return_intermediates:
        return j3, mesh, dict(g=g, vert0=vert0, v1=v1, v2=v2, v3=v3, j1=j1, j2=j2)
    return j3, mesh


def pmce_forward(sd, pose2d, img_feat, vj_relation, depth=3, dtype=torch.float32):
    """PMCE.forward (PMCE.py:15-20) -> (cam_mesh[B,6890,3] m, cam_pose[B,J,3] m, pose3d[B,J,3] mm)."""
    pose3d = lifter_forward(sd, pose2d, img_feat, depth, "pose_lifter.", dtype)
    pose3d = pose3d.reshape(-1, pose3d.shape[-2], 3)
    cam_pose, cam_mesh = decoder_forward(sd, pose3d / 1000, img_feat, vj_relation, "pose_mesh_coevo.", dtype)
    return cam_mesh, cam_pose, pose3d


def j_regress(mesh_m, j_regressor, dtype=torch.float32):
    """Caller-side tail (lib/core/base.py:223-225): pred_pose = J_regressor[None] @ (pred_mesh*1000)."""
    J = torch.as_tensor(np.asarray(j_regressor, dtype=np.float32)).to(dtype)     # torch.Tensor(float64 file) -> fp32
    return torch.matmul(J[None, :, :], mesh_m.to(dtype) * 1000)


def flops_per_clip(num_joint=17, embed_dim=256, depth=3):
    """Reference-equivalent FLOPs of one clip (2*MACs of every matmul as the reference computes them,
    dead code included) — SURVEY §8d; J=17,C=256 -> 3.55e9."""
    T, J, C, F_, D, V, VF, H = 16, num_joint, embed_dim, 2048, 64, 431, 6890, 1024
    tok = T * J
    lifter = 2 * T * F_ * C + 2 * tok * 2 * C
    per_blk = 2 * tok * C * 3 * C + 2 * tok * C * C + 2 * 2 * tok * C * 2 * C
    attn_s = T * 8 * 2 * 2 * J * J * (C // 8)
    attn_t = J * 8 * 2 * 2 * T * T * (C // 8)
    lifter += depth * (2 * per_blk + attn_s + attn_t) + 2 * tok * C * 3
    gru = 2 * (2 * T * 3 * H * F_ + 2 * T * 3 * H * H) * 2
    ada = 72 * 2 * F_ * D

    def ca(nq, nk):
        return 2 * nq * D * D * 2 + 2 * nk * D * D * 2 + 2 * 2 * nq * nk * D + 2 * 2 * nq * D * 4 * D

    def sa(n):
        return 2 * n * D * 3 * D + 2 * 2 * n * n * D + 2 * n * D * D + 2 * 2 * n * D * 4 * D

    blk = ca(J, V) + ca(V, J) + sa(J) + sa(V) + 2 * V * D * D + 2 * J * D * D + 2 * (J + V) * 3 * D * 2
    up = 2 * 3 * V * 3 * VF + 3 * 2 * 2 * H * VF
    return dict(lifter=lifter, gru=gru, coevo=3 * blk + ada, upsample=up, total=lifter + gru + 3 * blk + ada + up)
