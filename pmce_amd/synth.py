"""Deterministic synthetic weights, inputs and base-data stand-ins.

No pretrained checkpoint, dataset or SMPL asset is available (SURVEY §7 "Missing assets"), so every
test, fixture and benchmark regenerates a full reference-layout ``state_dict`` (412 MB fp32) from a
closed-form counter-based generator instead of committing it.  The generator is pure integer
arithmetic (splitmix64 over (seed, crc32(name), element index)) so it is bit-identical on every
machine, torch version and thread count.

Key layout = the reference checkpoint layout, SURVEY §8(b) (reference lib/models/PMCE.py:11-13,
PoseEstimation.py:31-66, CoevoDecoder.py:16-29,31-46,64-81,89-101,107-117,133-173,194-224);
tests/golden/make_golden.py loads it into the real reference modules with ``strict=True``.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import numpy as np

from .config import FEAT_DIM, GRU_HIDDEN, NUM_VERTS, NUM_VERTS_FULL, SEQLEN

_U64 = np.uint64


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = x + _U64(0x9E3779B97F4A7C15)
    z = x
    z = (z ^ (z >> _U64(30))) * _U64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> _U64(27))) * _U64(0x94D049BB133111EB)
    return z ^ (z >> _U64(31))


def uniform_pm1(name: str, n: int, seed: int) -> np.ndarray:
    """n float32 values in [-1, 1), a pure function of (name, seed, index)."""
    with np.errstate(over="ignore"):
        key = _U64((zlib.crc32(name.encode()) << 32) ^ (seed & 0xFFFFFFFF))
        base = _splitmix64(np.array([key], dtype=_U64))[0]
        idx = np.arange(n, dtype=_U64)
        z = _splitmix64(idx * _U64(0xD1342543DE82EF95) + base)
    u24 = (z >> _U64(40)).astype(np.float32)            # exact in fp32
    return u24 * np.float32(2.0 / (1 << 24)) - np.float32(1.0)


# ----------------------------------------------------------------------------------------------
# state_dict spec
# ----------------------------------------------------------------------------------------------

def _lin(spec, prefix, out_f, in_f, bias=True, scale=1.0):
    b = scale / np.sqrt(in_f)
    spec[prefix + ".weight"] = ((out_f, in_f), b, 0.0)
    if bias:
        spec[prefix + ".bias"] = ((out_f,), b, 0.0)


def _ln(spec, prefix, c):
    spec[prefix + ".weight"] = ((c,), 0.1, 1.0)   # 1 + 0.1 u
    spec[prefix + ".bias"] = ((c,), 0.1, 0.0)


def lifter_spec(num_joint: int, embed_dim: int, depth: int, prefix: str = "") -> "OrderedDict[str, tuple]":
    """name -> (shape, uniform half-width, offset) for GraphormerNet (PoseEstimation.py:31-66)."""
    C = embed_dim
    s: OrderedDict = OrderedDict()
    s[prefix + "spatial_pos_embed"] = ((1, num_joint, C), 0.1, 0.0)
    s[prefix + "temporal_pos_embed"] = ((1, SEQLEN, C), 0.1, 0.0)
    _lin(s, prefix + "joint_embed", C, 2)
    _lin(s, prefix + "imgfeat_embed", C, FEAT_DIM)
    for kind in ("SpatialBlocks", "TemporalBlocks"):
        for i in range(depth):
            p = f"{prefix}{kind}.{i}"
            _ln(s, p + ".norm1", C)
            _lin(s, p + ".attn.qkv", 3 * C, C)
            _lin(s, p + ".attn.proj", C, C)
            _ln(s, p + ".norm2", C)
            _lin(s, p + ".mlp.fc1", 2 * C, C)      # mlp_ratio = 2 (PoseEstimation.py:32)
            _lin(s, p + ".mlp.fc2", C, 2 * C)
    _ln(s, prefix + "norm_s", C)
    _ln(s, prefix + "norm_t", C)
    _ln(s, prefix + "regression.0", C)
    # x1000: trained-scale realism, pose3d comes out in millimetres (~1e2..1e3), SURVEY §8c
    _lin(s, prefix + "regression.1", 3, C, scale=1000.0)
    s[prefix + "fusion.weight"] = ((1, SEQLEN, 1, 1), 0.25, 0.0)
    s[prefix + "fusion.bias"] = ((1,), 0.25, 0.0)
    return s


def _adaln(s, p, d):
    _lin(s, p + ".mlp_gamma", d, FEAT_DIM)
    _lin(s, p + ".mlp_beta", d, FEAT_DIM)


def _coevo_block_spec(s, p, J, V, D):
    sq3 = float(np.sqrt(3.0))                       # unit-variance uniform (reference uses randn)
    _lin(s, p + ".joint_proj", D, 3)
    _lin(s, p + ".vertx_proj", D, 3)
    # registration order of the reference module (CoevoDecoder.py:148-173) is irrelevant to
    # load_state_dict; we keep a readable order.
    s[p + ".joint_pos_embed"] = ((1, J, D), sq3, 0.0)
    s[p + ".vertx_pos_embed"] = ((1, V, D), sq3, 0.0)
    s[p + ".j_Q_embed"] = ((1, J, D), sq3, 0.0)
    s[p + ".v_Q_embed"] = ((1, V, D), sq3, 0.0)
    _lin(s, p + ".proj_v2j_dim", D, D)
    _lin(s, p + ".proj_j2v_dim", D, D)
    s[p + ".v2j_K_embed"] = ((1, V, D), sq3, 0.0)
    s[p + ".j2v_K_embed"] = ((1, J, D), sq3, 0.0)
    for stream in ("joint", "vertx"):
        q = f"{p}.{stream}_SA_FFN"
        _adaln(s, q + ".norm1", D)
        _lin(s, q + ".attn.qkv", 3 * D, D)
        _lin(s, q + ".attn.proj", D, D)
        _adaln(s, q + ".norm2", D)
        _lin(s, q + ".mlp.fc1", 4 * D, D)
        _lin(s, q + ".mlp.fc2", D, 4 * D)
    for stream in ("joint", "vertx"):
        q = f"{p}.{stream}_CA_FFN"
        for n in ("normq", "normk", "normv"):
            _adaln(s, f"{q}.{n}", D)
        for w in ("wq", "wk", "wv", "proj"):
            _lin(s, f"{q}.attn.{w}", D, D)
        _adaln(s, q + ".norm2", D)
        _lin(s, q + ".mlp.fc1", 4 * D, D)
        _lin(s, q + ".mlp.fc2", D, 4 * D)
    _lin(s, p + ".proj_joint_feat2coor", 3, D)
    _lin(s, p + ".proj_vertx_feat2coor", 3, D)


def decoder_spec(num_joint: int, joint_dim: int = 64, prefix: str = "") -> "OrderedDict[str, tuple]":
    """name -> (shape, half-width, offset) for Pose2Mesh (CoevoDecoder.py:194-224)."""
    s: OrderedDict = OrderedDict()
    s[prefix + "init_vertices"] = ((NUM_VERTS, 3), 0.5, 0.0)   # buffer; not read by forward
    for b in (1, 2, 3):
        _coevo_block_spec(s, f"{prefix}coevoblock{b}", num_joint, NUM_VERTS, joint_dim)
    fan = NUM_VERTS * 3
    s[prefix + "upsample_conv.weight"] = ((NUM_VERTS_FULL, NUM_VERTS, 3), 1.0 / np.sqrt(fan), 0.0)
    s[prefix + "upsample_conv.bias"] = ((NUM_VERTS_FULL,), 1.0 / np.sqrt(fan), 0.0)
    H = GRU_HIDDEN
    gb = 1.0 / np.sqrt(H)
    for layer in (0, 1):
        for sfx in ("", "_reverse"):
            in_f = FEAT_DIM if layer == 0 else 2 * H
            s[f"{prefix}gru_cur.weight_ih_l{layer}{sfx}"] = ((3 * H, in_f), gb, 0.0)
            s[f"{prefix}gru_cur.weight_hh_l{layer}{sfx}"] = ((3 * H, H), gb, 0.0)
            s[f"{prefix}gru_cur.bias_ih_l{layer}{sfx}"] = ((3 * H,), gb, 0.0)
            s[f"{prefix}gru_cur.bias_hh_l{layer}{sfx}"] = ((3 * H,), gb, 0.0)
    for i in (1, 2, 3):
        _lin(s, f"{prefix}linear_cur{i}", NUM_VERTS_FULL, 2 * H)
    return s


def pmce_spec(num_joint: int, embed_dim: int = 256, depth: int = 3) -> "OrderedDict[str, tuple]":
    """Full PMCE checkpoint layout (PMCE.py:11-13): pose_lifter.* + pose_mesh_coevo.*"""
    s = lifter_spec(num_joint, embed_dim, depth, "pose_lifter.")
    s.update(decoder_spec(num_joint, 64, "pose_mesh_coevo."))
    return s


def make_state_dict(spec, seed: int = 123, as_torch: bool = True):
    """Materialise a spec. seed 123 = the reference's default --seed (main/test.py:12)."""
    out = OrderedDict()
    for name, (shape, half, off) in spec.items():
        n = int(np.prod(shape))
        v = uniform_pm1(name, n, seed) * np.float32(half) + np.float32(off)
        out[name] = v.reshape(shape)
    if as_torch:
        import torch
        out = OrderedDict((k, torch.from_numpy(v)) for k, v in out.items())
    return out


# ----------------------------------------------------------------------------------------------
# synthetic inputs and base-data stand-ins (SURVEY §8c shim 6)
# ----------------------------------------------------------------------------------------------

def make_inputs(batch: int, num_joint: int, seed: int = 0):
    """pose2d[B,16,J,2] ~ U(-1,1) (normalize_screen_coordinates range, PW3D/dataset.py:202-204);
    img_feat[B,16,2048] >= 0 and sparse-ish like ResNet avg-pool features: max(0, 1.5*u - 0.3)."""
    p = uniform_pm1("input.pose2d", batch * SEQLEN * num_joint * 2, seed)
    f = uniform_pm1("input.img_feat", batch * SEQLEN * FEAT_DIM, seed + 1)
    f = np.maximum(np.float32(1.5) * f - np.float32(0.3), np.float32(0.0))
    return (p.reshape(batch, SEQLEN, num_joint, 2).copy(), f.reshape(batch, SEQLEN, FEAT_DIM).copy())


def make_decoder_inputs(batch: int, num_joint: int, seed: int = 0):
    """joints[B,J,3] ~ 0.5*U(-1,1) metres (BASELINE configs[1]: decoder-only forward)."""
    j = uniform_pm1("input.joints", batch * num_joint * 3, seed) * np.float32(0.5)
    _, f = make_inputs(batch, num_joint, seed)
    return j.reshape(batch, num_joint, 3).copy(), f


def make_base_data(seed: int = 7):
    """Stand-ins for the absent SMPL-derived assets: a smooth-ish random mean mesh [6890,3] (float32,
    like smpl_mean_vertices.npy) and the two down-sampling matrices of mesh_downsampling.npz as
    averaging-selection CSR matrices D0[1723,6890], D1[431,1723] (scipy sparse, float32)."""
    import scipy.sparse as sp
    v = uniform_pm1("base.mean_vertices", NUM_VERTS_FULL * 3, seed).reshape(NUM_VERTS_FULL, 3)
    v = (v * np.array([0.35, 0.9, 0.15], dtype=np.float32)).astype(np.float32)   # body-like extent (m)

    def sel(n_out, n_in, tag):
        # each coarse vertex = 0.6*v[a] + 0.4*v[b]  (real D matrices are sparse quadric-decimation maps)
        a = (np.arange(n_out) * n_in) // n_out
        r = uniform_pm1(tag, n_out, seed)
        b = np.minimum(n_in - 1, a + 1 + ((r + 1.0) * 1.4).astype(np.int64))
        rows = np.repeat(np.arange(n_out), 2)
        cols = np.stack([a, b], 1).reshape(-1)
        vals = np.tile(np.array([0.6, 0.4], dtype=np.float32), n_out)
        return sp.csr_matrix((vals, (rows, cols)), shape=(n_out, n_in), dtype=np.float32)

    return v, [sel(1723, NUM_VERTS_FULL, "base.D0"), sel(NUM_VERTS, 1723, "base.D1")]
