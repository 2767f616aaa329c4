"""Input staging: oracle vs the reference's own functions (golden), host window table vs both."""
import os.path as osp
import sys

import numpy as np
import pytest

HERE = osp.dirname(osp.abspath(__file__))
sys.path.insert(0, osp.dirname(HERE))
sys.path.insert(0, osp.join(HERE, "golden"))
from oracle import staging_oracle as so  # noqa: E402
from pmce_amd import staging  # noqa: E402
import make_golden_staging as mg  # noqa: E402  (only its input generators; the reference is not touched)

G = np.load(osp.join(HERE, "golden", "staging.npz"))


def test_oracle_pelvis_neck_matches_reference():
    kp, _ = mg.inputs()
    ext = np.stack([so.add_pelvis_and_neck(k) for k in kp])
    np.testing.assert_array_equal(ext, G["ext"])
    np.testing.assert_array_equal(np.stack([so.add_pelvis_and_neck(k, only_pelvis=True) for k in kp]), G["ext_pelvis"])


def test_oracle_normalisation_matches_reference():
    kp, shapes = mg.inputs()
    norm = np.stack([np.array(so.normalize_screen_coordinates(so.add_pelvis_and_neck(kp[i])[:, :2], w=shapes[i][1], h=shapes[i][0]),
                              dtype=np.float32) for i in range(len(kp))])
    np.testing.assert_allclose(norm, G["norm"], rtol=0, atol=2e-7)   # numpy-version float promotion differs by <= 1 ulp


@pytest.mark.parametrize("tag,seqlen,stride,mv", [("s1", 16, 1, True), ("s16", 16, 16, True), ("s4", 16, 4, True), ("s1_nov", 16, 1, False)])
def test_window_tables_match_reference(tag, seqlen, stride, mv):
    names, valid = mg.video_layout()
    want = G["win_" + tag]
    np.testing.assert_array_equal(so.split_into_chunks_mesh(names, seqlen, stride, valid, match_vibe=mv), want)
    np.testing.assert_array_equal(staging.mesh_window_table(names, seqlen, stride, valid, match_vibe=mv), want)


def test_window_table_edge_cases():
    assert staging.mesh_window_table([], 16, 1).shape == (0, 2)
    short = [f"0/s/image_{i:05d}.jpg" for i in range(15)]
    assert staging.mesh_window_table(short, 16, 1).shape == (0, 2)
    exact = [f"0/s/image_{i:05d}.jpg" for i in range(16)]
    np.testing.assert_array_equal(staging.mesh_window_table(exact, 16, 1), [[0, 15]])
    np.testing.assert_array_equal(staging.mesh_window_table(exact, 16, 1, mid_valid=np.arange(16) != 8), np.zeros((0, 2)))


@pytest.mark.parametrize("tag,seqlen,stride,mv", [("s1", 16, 1, True), ("s16", 16, 16, True), ("s4", 16, 4, True), ("s1_nov", 16, 1, False)])
def test_pose_window_tables_match_reference(tag, seqlen, stride, mv):
    """split_into_chunks_pose of the REAL reference (fixture) vs the oracle restatement and the product's host function."""
    names, _ = mg.video_layout()
    want = G["pose_win_" + tag]
    np.testing.assert_array_equal(so.split_into_chunks_pose(names, seqlen, stride, match_vibe=mv), want)
    np.testing.assert_array_equal(staging.pose_window_table(names, seqlen, stride, match_vibe=mv), want)


def test_single_video_windows_match_reference():
    """pmce_amd.streaming.window_indices (one video, stride 1) vs the reference's split_into_chunks_pose on single videos of
    several lengths, including the ones around the 16-frame VIBE chunk boundaries."""
    from pmce_amd import streaming
    for L in mg.SINGLE_VIDEO_LENGTHS:
        want = G[f"pose_win_single_{L}"]
        np.testing.assert_array_equal(streaming.window_indices(L), want)
        one = [f"0/only/image_{i:05d}.jpg" for i in range(L)]
        np.testing.assert_array_equal(staging.pose_window_table(one), want)
        np.testing.assert_array_equal(so.split_into_chunks_pose(one, 16, 1), want)
