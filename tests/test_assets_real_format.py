"""The REAL-asset branch of the init-time work: ``assets.build_template`` reading a ``data/base_data`` written in
the reference's own format - object arrays of scipy sparse matrices, weighted multi-entry ``D``, vertices planted on the
bisector planes of template joints - against what the reference's own ``Mesh.downsample`` / ``build_verts_joints_relation``
produced from the same files (tests/golden/make_golden_assets.py; lib/models/backbones/mesh.py:59-96,
graph_layers.py:12-30, graph_utils.py:27-46, CoevoDecoder.py:197-209)."""
import os.path as osp

import numpy as np
import pytest
import scipy.sparse as sp

from pmce_amd import assets

GOLD = osp.join(osp.dirname(osp.abspath(__file__)), "golden", "assets_real_format.npz")


def write_base_data(z, base_dir):
    """The two files as the reference reads them (mesh.py:49-55: np.load(..., encoding='latin1', allow_pickle=True))."""
    np.save(osp.join(base_dir, "smpl_mean_vertices.npy"), z["mean_vertices"])
    d0 = sp.csc_matrix((z["d0_data"], z["d0_indices"], z["d0_indptr"]), shape=(1723, 6890))
    d1 = sp.coo_matrix((z["d1_data"], (z["d1_row"], z["d1_col"])), shape=(431, 1723))
    assert str(z["d0_format"]) == "csc" and str(z["d1_format"]) == "coo"
    tiny = sp.identity(4, format="csr", dtype=np.float32)
    A = np.empty(3, dtype=object)
    U = np.empty(2, dtype=object)
    D = np.empty(2, dtype=object)
    for i in range(3):
        A[i] = tiny
    U[0] = U[1] = tiny
    D[0], D[1] = d0, d1
    np.savez(osp.join(base_dir, "mesh_downsampling.npz"), A=A, U=U, D=D)
    return d0, d1


def test_files_branch_matches_the_reference_bit_for_bit(tmp_path):
    z = np.load(GOLD)
    d0, d1 = write_base_data(z, str(tmp_path))
    # the fixture is not a toy: most rows of D carry several unequal weights, and vertices sit on joint bisectors
    assert (np.diff(d0.tocsr().indptr) > 1).sum() > 1000 and (np.diff(sp.csr_matrix(d1).indptr) > 1).sum() > 300
    assert int(z["near_ties"]) >= 12
    assets.allow_synthetic_base_data(False)
    try:
        v431, vj, src = assets.build_template(base_dir=str(tmp_path))
    finally:
        assets.allow_synthetic_base_data(True)
    assert src == "files"
    assert np.array_equal(vj, z["vj_relation"])                 # integer table: bit-exact, ties and near-ties included
    assert vj.dtype == np.int64 and vj.min() >= 0 and vj.max() <= 16
    assert np.array_equal(v431, z["init_vertices"])             # same torch call on the same COO entries: the same bits
    assert np.array_equal(assets.template_joints(z["mean_vertices"]), z["joints_template"])


def test_bit_equality_is_not_vacuous(tmp_path):
    """A summation order other than the reference's (scipy's ``csr @ x`` - what this repo used before round 4) gives the same
    map with other last bits on hundreds of template coordinates, and the fixture holds vertices whose two nearest joints are
    ulps apart: equal bits of the template and of the template joints are what makes ``vj_relation`` the reference's."""
    z = np.load(GOLD)
    d0, d1 = write_base_data(z, str(tmp_path))
    x = z["mean_vertices"].astype(np.float32)
    alt = sp.csr_matrix(d1).astype(np.float32) @ (sp.csr_matrix(d0).astype(np.float32) @ x)
    assert np.abs(alt - z["init_vertices"]).max() < 1e-6
    assert (alt != z["init_vertices"]).sum() > 100
    jt = z["joints_template"]
    d = ((z["init_vertices"][:, None, :] - jt[None]) ** 2).sum(-1)
    s = np.sort(d, axis=1)
    gap = (s[:, 1] - s[:, 0]) / s[:, 0]
    assert (gap < 1e-6).sum() >= 12


def test_missing_files_raise_without_the_opt_in(tmp_path):
    assets.allow_synthetic_base_data(False)
    try:
        import os
        old = os.environ.pop("PMCE_SYNTHETIC_BASE_DATA", None)
        try:
            with pytest.raises(FileNotFoundError):
                assets.build_template(base_dir=str(tmp_path))
        finally:
            if old is not None:
                os.environ["PMCE_SYNTHETIC_BASE_DATA"] = old
    finally:
        assets.allow_synthetic_base_data(True)
