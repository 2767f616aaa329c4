"""Init-time host assets of the path: joint regressors, the 431-vertex template and ``vj_relation``.

Mirrors the one-time work of ``Pose2Mesh.__init__`` (reference lib/models/CoevoDecoder.py:197-209):
mean mesh --D0--> 1723 --D1--> 431 vertices (lib/models/backbones/mesh.py:81-96, graph_layers.py:29),
template joints = J_regressor_h36m @ mean mesh (CoevoDecoder.py:207-208) and the per-vertex nearest
joint table (lib/graph_utils.py:27-46).  This is init-time host code (numpy); the per-clip gather
that consumes the table runs on the GPU.
"""
from __future__ import annotations

import os
import os.path as osp

import numpy as np

from .config import NUM_VERTS, NUM_VERTS_FULL, cfg

_DATA = osp.join(osp.dirname(osp.abspath(__file__)), "data")


def load_j_regressor(name: str = "h36m") -> np.ndarray:
    """Dense float64 [17,6890] regressor ('h36m' = J_regressor_h36m_correct.npy, 107 nnz;
    'coco' = J_regressor_coco.npy, 105 nnz), from the bundled CSR copy of the reference's data
    files (loaded by the reference at lib/smpl.py:18-19, CoevoDecoder.py:207)."""
    z = np.load(osp.join(_DATA, "j_regressors_csr.npz"))
    shape = tuple(int(v) for v in z[name + "_shape"])
    out = np.zeros(shape, dtype=np.float64)
    indptr, indices, data = z[name + "_indptr"], z[name + "_indices"], z[name + "_data"]
    for r in range(shape[0]):
        out[r, indices[indptr[r]:indptr[r + 1]]] = data[indptr[r]:indptr[r + 1]]
    return out


def regressor_to_csr(reg: np.ndarray):
    """Dense [R,6890] -> (indptr int32[R+1], indices int32[nnz], data float32[nnz]).
    The caller casts the float64 file to fp32 (``torch.Tensor(...)``, lib/core/base.py:196)."""
    reg = np.asarray(reg)
    indptr = [0]
    indices, data = [], []
    for r in range(reg.shape[0]):
        nz = np.nonzero(reg[r])[0]
        indices.extend(nz.tolist())
        data.extend(reg[r, nz].astype(np.float32).tolist())
        indptr.append(len(indices))
    return (np.asarray(indptr, np.int32), np.asarray(indices, np.int32), np.asarray(data, np.float32))


def _torch_coo(d):
    """scipy sparse -> torch COO exactly as the reference builds it (mesh.py:16-21: ``coo_matrix(D)``, entries in that
    order, fp32 values, NOT coalesced), so that the product below adds a row's terms in the reference's order."""
    import scipy.sparse as sp
    import torch
    c = sp.coo_matrix(d)
    idx = torch.from_numpy(np.array([c.row, c.col]).astype(np.int64))
    val = torch.from_numpy(np.asarray(c.data).astype(np.float32))
    return torch.sparse_coo_tensor(idx, val, c.shape)


def downsample_template(mean_vertices: np.ndarray, D) -> np.ndarray:
    """[6890,3] f32 -> [431,3] f32 by the two sparse down-sampling maps.  The reference runs ``torch.matmul(sparse COO,
    dense)`` in fp32 (mesh.py:81-96 via graph_layers.py:19,29); the same torch call on the same COO entries is used here, so
    the template - and with it every near-tie ``argmin`` of ``vj_relation`` - has the bits of the reference code EXECUTED ON CPU (a
    scipy ``csr @ x`` adds a multi-entry row's terms in another order: last-ulp differences, which can flip a tie).  The reference
    itself runs these products on CUDA (mesh.py:62 ``device=cuda``, CoevoDecoder.py:200,207 ``.cuda()``): cuSPARSE / cuBLAS summation
    order is not reproducible here, so for a vertex whose two nearest joints tie to the last ulp the table may differ from the one a
    real run built - a caller who has that run's ``vj_relation`` can pass it (``model.vj_relation = ...`` before ``.to(device)``).  The
    golden test ties bit-equality to the installed torch's CPU sparse matmul (tests/test_assets_real_format.py says so)."""
    import torch
    x = torch.from_numpy(np.ascontiguousarray(mean_vertices, dtype=np.float32))
    for d in D[:2]:
        x = torch.matmul(_torch_coo(d), x)
    x = x.numpy()
    assert x.shape == (NUM_VERTS, 3), x.shape
    return x


def template_joints(mean_vertices: np.ndarray) -> np.ndarray:
    """[17,3] f32 = J_regressor_h36m (cast to fp32) @ mean mesh as ONE dense fp32 ``torch.matmul``
    (CoevoDecoder.py:207-208) - the reference's call, hence its summation order."""
    import torch
    jreg = torch.from_numpy(load_j_regressor("h36m").astype(np.float32))
    return torch.matmul(jreg, torch.from_numpy(np.ascontiguousarray(mean_vertices, dtype=np.float32))).numpy()


def build_verts_joints_relation(joints: np.ndarray, vertices: np.ndarray) -> np.ndarray:
    """Nearest template joint of every vertex (graph_utils.py:27-46): argmin over joints of the squared
    distance, first index on ties (numpy argmin).  Returns int64[V] (the reference keeps the ints
    in a float64 ndarray and indexes with it; values identical)."""
    joints = np.asarray(joints)
    vertices = np.asarray(vertices)
    d = ((vertices[:, None, :] - joints[None, :, :]) ** 2).sum(-1)   # same op order: sub, square, sum(axis=xyz)
    return np.argmin(d, axis=1).astype(np.int64)


_ALLOW_SYNTHETIC = [False]


def allow_synthetic_base_data(enable: bool = True):
    """Opt in to the synthetic stand-ins for the SMPL-derived base data (tests, benchmarks and fixtures call this; so does
    ``PMCE_SYNTHETIC_BASE_DATA=1``).  Without the opt-in a missing ``smpl_mean_vertices.npy`` / ``mesh_downsampling.npz``
    raises, as the reference's ``np.load`` would (CoevoDecoder.py:194, mesh.py:59): ``vj_relation`` is derived from these
    files and is not in the checkpoint, so a real checkpoint on a synthetic template gives plausible but wrong meshes."""
    _ALLOW_SYNTHETIC[0] = bool(enable)


def synthetic_base_data_allowed() -> bool:
    return _ALLOW_SYNTHETIC[0] or os.environ.get("PMCE_SYNTHETIC_BASE_DATA", "") not in ("", "0")


def load_base_data(base_dir: str | None = None):
    """(mean_vertices[6890,3] f32, [D0, D1], source) from ``cfg.DATASET.BASE_DATA_DIR``: the user-supplied SMPL-derived
    files (smpl_mean_vertices.npy, mesh_downsampling.npz — CoevoDecoder.py:194, mesh.py:59).  When they are missing:
    FileNotFoundError, unless the caller opted in to the synthetic stand-ins of :func:`pmce_amd.synth.make_base_data`
    (:func:`allow_synthetic_base_data`)."""
    base_dir = base_dir or cfg.DATASET.BASE_DATA_DIR
    mv = osp.join(base_dir, "smpl_mean_vertices.npy")
    md = osp.join(base_dir, "mesh_downsampling.npz")
    if osp.exists(mv) and osp.exists(md):
        import scipy.sparse as sp
        v = np.load(mv).astype(np.float32)
        z = np.load(md, encoding="latin1", allow_pickle=True)
        D = [sp.coo_matrix(d) for d in z["D"][:2]]   # entry order kept: it is the order the reference's spmm adds in
        return v, D, "files"
    if not synthetic_base_data_allowed():
        raise FileNotFoundError(
            f"{mv} / {md} not found.  vj_relation (which joint every mesh vertex starts from) is derived from these "
            "SMPL-derived files and is not stored in checkpoints; set cfg.DATASET.BASE_DATA_DIR (env PMCE_BASE_DATA_DIR) to "
            "the reference's data/base_data.  Tests and benchmarks on synthetic weights opt in to a synthetic template with "
            "pmce_amd.assets.allow_synthetic_base_data() or PMCE_SYNTHETIC_BASE_DATA=1.")
    from .synth import make_base_data
    v, D = make_base_data()
    return v, D, "synthetic"


def build_template(base_dir: str | None = None):
    """Everything ``Pose2Mesh.__init__`` derives from base data: (init_vertices[431,3] f32,
    vj_relation int64[431] in 0..16, source tag)."""
    v, D, src = load_base_data(base_dir)
    assert v.shape == (NUM_VERTS_FULL, 3)
    v431 = downsample_template(v, D)
    vj = build_verts_joints_relation(template_joints(v), v431)
    return v431, vj, src
