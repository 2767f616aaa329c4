#!/usr/bin/env python3
"""Fixture for the REAL-asset branch of the init-time work (VERDICT r03 missing #4 / weak #5).

Writes a ``data/base_data`` directory in the reference's own format - ``smpl_mean_vertices.npy`` and
``mesh_downsampling.npz`` with OBJECT arrays ``A``, ``U``, ``D`` of scipy sparse matrices (what
``np.load(..., encoding='latin1', allow_pickle=True)`` reads at lib/models/backbones/mesh.py:49-55) - whose ``D`` are
weighted multi-entry maps like the quadric-decimation matrices of the real file (1-6 entries per row, unequal weights,
one stored column-major (CSC), one as COO in shuffled entry order), with vertices planted ON the bisector planes between
template joints (exact ties and +-1 ulp near-ties of the ``argmin``).  Then runs the reference's OWN code on it -
``Mesh()`` + ``Mesh.downsample`` (mesh.py:59-96, ``spmm`` graph_layers.py:12-30), ``torch.matmul(J_regressor, init_vertices)``
(CoevoDecoder.py:207-208) and ``build_verts_joints_relation`` (graph_utils.py:27-46) - and stores

  inputs   mean_vertices[6890,3] f32 ; D0 / D1 as (format tag, data, row/indices, col/indptr) arrays
  outputs  init_vertices[431,3] f32, joints_template[17,3] f32, vj_relation int64[431]

in tests/golden/assets_real_format.npz.  tests/test_assets_real_format.py rewrites the files from those arrays and drives
``pmce_amd.assets.build_template(base_dir=...)`` through its ``"files"`` branch.  Data only; nothing of the reference is copied.
Runs only in the build container (needs /root/reference).
"""
import os
import os.path as osp
import sys
import tempfile

import numpy as np
import scipy.sparse as sp
import torch

HERE = osp.dirname(osp.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, osp.dirname(osp.dirname(HERE)))
import make_golden as MG  # noqa: E402  (the import shims)
from pmce_amd import assets, synth  # noqa: E402

NV, N1, N2 = 6890, 1723, 431


def decimation_like(n_out, n_in, rng, onehot_rows=()):
    """Rows with 1-6 entries, positive unequal weights summing to ~1 (not exactly: the real maps are not exactly
    stochastic in fp32 either); `onehot_rows` are forced to ONE entry of weight 1.0 (a kept vertex)."""
    rows, cols, vals = [], [], []
    for r in range(n_out):
        base = (r * n_in) // n_out
        if r in onehot_rows:
            rows.append(r); cols.append(base); vals.append(1.0)
            continue
        k = int(rng.integers(1, 7))
        c = np.unique(np.clip(base + rng.integers(-40, 41, size=k), 0, n_in - 1))
        w = rng.random(len(c)) ** 2 + 0.05
        w = (w / w.sum()).astype(np.float32)
        rows += [r] * len(c); cols += c.tolist(); vals += w.tolist()
    return sp.coo_matrix((np.asarray(vals, np.float32), (np.asarray(rows), np.asarray(cols))), shape=(n_out, n_in))


def main():
    MG.install_shims()
    rng = np.random.default_rng(20260927)
    v, _ = synth.make_base_data()
    v = v.copy()
    jreg = assets.load_j_regressor("h36m").astype(np.float32)
    used = set(np.nonzero(jreg.any(0))[0].tolist())     # vertices the template joints depend on: left alone

    # planted vertices: coarse vertex p <- D1 one-hot <- D0 one-hot <- one fine vertex we are free to place
    planted = list(range(5, N2, 17))[:24]
    d1 = decimation_like(N2, N1, rng, onehot_rows=set(planted))
    mids = [int((d1.tocsr()[p].indices)[0]) for p in planted]
    d0 = decimation_like(N1, NV, rng, onehot_rows=set(mids))
    fines = [int((d0.tocsr()[m].indices)[0]) for m in mids]
    assert not (set(fines) & used), "a planted vertex feeds the joint regressor: pick other rows"
    jt = torch.matmul(torch.from_numpy(jreg), torch.from_numpy(v)).numpy()       # template joints (do not depend on `fines`)
    def dists(p):   # the reference's arithmetic (graph_utils.py:37-40): fp32 sub, square, sum over xyz
        return ((p[None, :] - jt) ** 2).sum(1)

    for i, f in enumerate(fines):
        a = int(rng.integers(0, 17))
        b = int(np.argsort(((jt - jt[a]) ** 2).sum(1))[1])          # a's nearest neighbour: their midpoint has a, b as its two nearest joints
        mid = ((jt[a].astype(np.float64) + jt[b].astype(np.float64)) / 2).astype(np.float32)
        cands = [mid]
        for ax in range(3):                                           # the fp32 neighbours of the midpoint along each axis
            for sgn in (np.inf, -np.inf):
                c = mid.copy(); c[ax] = np.nextafter(c[ax], np.float32(sgn)); cands.append(c)
        want_tie = i % 3 == 0
        best = None
        for c in cands:
            d = dists(c)
            o = np.argsort(d)
            if set(o[:2].tolist()) != {a, b}:
                continue
            tie = d[a] == d[b]
            if tie == want_tie:
                best = c
                break
            best = best if best is not None else c
        v[f] = best if best is not None else mid
    jt2 = torch.matmul(torch.from_numpy(jreg), torch.from_numpy(v)).numpy()
    assert np.array_equal(jt, jt2)

    # the files, in the reference's format: D0 column-major (CSC), D1 COO with shuffled entries
    d0_file = d0.tocsc()
    perm = rng.permutation(d1.nnz)
    d1_file = sp.coo_matrix((d1.data[perm], (d1.row[perm], d1.col[perm])), shape=d1.shape)
    cwd = tempfile.mkdtemp(prefix="pmce_assets_")
    base = osp.join(cwd, "data", "base_data")
    os.makedirs(base)
    os.makedirs(osp.join(cwd, "data", "Human36M"))
    np.save(osp.join(base, "smpl_mean_vertices.npy"), v)
    tiny = sp.identity(4, format="csr", dtype=np.float32)
    A = np.empty(3, dtype=object); U = np.empty(2, dtype=object); D = np.empty(2, dtype=object)
    for i in range(3):
        A[i] = tiny
    U[0] = U[1] = tiny
    D[0], D[1] = d0_file, d1_file
    np.savez(osp.join(base, "mesh_downsampling.npz"), A=A, U=U, D=D)
    np.save(osp.join(cwd, "data", "Human36M", "J_regressor_h36m_correct.npy"),
            np.load(osp.join(MG.REF, "data", "Human36M", "J_regressor_h36m_correct.npy")))
    os.chdir(cwd)

    # ---- the reference's own init code (Pose2Mesh.__init__ lines 197-209, piece by piece) ----
    from models.backbones.mesh import Mesh
    import graph_utils
    mesh = Mesh()
    init_vertices = torch.from_numpy(np.load(osp.join("data", "base_data", "smpl_mean_vertices.npy"))).cuda()
    v1723 = mesh.downsample(init_vertices)
    v431 = mesh.downsample(v1723, n1=1, n2=2)
    J_regressor = torch.from_numpy(np.load("data/Human36M/J_regressor_h36m_correct.npy").astype(np.float32)).cuda()
    joints_template = torch.matmul(J_regressor, init_vertices)
    vj, _ = graph_utils.build_verts_joints_relation(joints_template.cpu().numpy(), v431.cpu().numpy())
    vj = np.asarray(vj).astype(np.int64)

    # how tight the planted ties are (reported, and stored so the test can assert the fixture is not trivial)
    d = ((v431.numpy()[:, None, :] - joints_template.numpy()[None]) ** 2).sum(-1)
    srt = np.sort(d, axis=1)
    gap = (srt[:, 1] - srt[:, 0]) / srt[:, 0]
    print("rows of D0 / D1 with > 1 entry:", int((np.diff(d0.tocsr().indptr) > 1).sum()), int((np.diff(d1.tocsr().indptr) > 1).sum()))
    print("vertices with relative gap between nearest and second joint < 1e-6:", int((gap < 1e-6).sum()), " exact ties:", int((gap == 0).sum()))

    np.savez_compressed(
        osp.join(HERE, "assets_real_format.npz"),
        mean_vertices=v,
        d0_format="csc", d0_data=d0_file.data, d0_indices=d0_file.indices.astype(np.int32), d0_indptr=d0_file.indptr.astype(np.int32),
        d1_format="coo", d1_data=d1_file.data, d1_row=d1_file.row.astype(np.int32), d1_col=d1_file.col.astype(np.int32),
        init_vertices=v431.numpy(), joints_template=joints_template.numpy(), vj_relation=vj,
        planted=np.asarray(planted, np.int32), near_ties=int((gap < 1e-6).sum()), exact_ties=int((gap == 0).sum()))
    print("wrote assets_real_format.npz")


if __name__ == "__main__":
    main()
