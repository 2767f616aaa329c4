"""pmce_amd/datasets.py (the reader of the reference's precomputed-feature / detection files) against the reference's
OWN loader: tests/golden/datasets_pw3d.npz holds what ``PW3D('test')`` (data/PW3D/dataset.py) made of the small synthetic directory
that tests/golden/pw3d_files.py writes (make_golden_datasets.py); the same directory is regenerated here and read by this package."""
import os.path as osp
import sys

import numpy as np
import pytest

HERE = osp.dirname(osp.abspath(__file__))
sys.path.insert(0, osp.join(HERE, "golden"))


@pytest.fixture(scope="module")
def table(tmp_path_factory):
    import pw3d_files
    from pmce_amd import datasets
    root = str(tmp_path_factory.mktemp("pw3d"))
    return datasets.load_pw3d(pw3d_files.write(root), "test")


@pytest.fixture(scope="module")
def gold():
    return np.load(osp.join(HERE, "golden", "datasets_pw3d.npz"))


def test_frame_tables_equal_the_reference_loader(table, gold):
    """Sorted by image path, the annotation without a feature skipped, every per-frame array the reference's load_data returns."""
    assert table.skipped == 1 and len(table) == len(gold["img_paths"]) == 134
    assert list(table.img_paths) == list(gold["img_paths"]) and list(table.vid_names) == list(gold["vid_names"])
    assert np.array_equal(table.img_shapes, gold["img_shapes"])
    assert np.array_equal(table.features[:, ::64], gold["features_sub"])
    assert np.array_equal(table.joints_cam_h36m, gold["joints_cam_h36m"]) and np.array_equal(table.joints_cam_coco, gold["joints_cam_coco"])
    # the reference stores the detector keypoints with pelvis and neck already appended ([N,19,3]): rows 0..16 are the file's
    assert np.array_equal(table.keypoints, gold["pred_pose2ds"][:, :17])


def test_window_list_and_model_inputs_equal_the_reference_dataset(table, gold):
    """vid_indices (split_into_chunks_mesh) and, for sampled windows, what __getitem__ hands the model - here through the host-side
    restatement of the staging arithmetic (oracle/staging_oracle.py; the GPU kernel is compared in test_gpu_staging.py)."""
    from oracle import staging_oracle as S
    from pmce_amd import datasets
    win = table.windows(16, 1)
    assert np.array_equal(win, gold["vid_indices"]) and len(win) == int(gold["n_items"])
    ext = np.stack([S.add_pelvis_and_neck(k) for k in table.keypoints])                                         # [N,19,3]
    assert np.array_equal(ext, gold["pred_pose2ds"])
    pose2d = np.stack([np.asarray(S.normalize_screen_coordinates(ext[i][:, :2], w=table.img_shapes[i][1], h=table.img_shapes[i][0]), dtype=np.float32)
                       for i in range(len(table))])
    gt = table.gt_joints_root_relative()
    frames = datasets.window_frames(win)
    for k in gold["sample_windows"]:
        idx = frames[k]
        assert np.array_equal(pose2d[idx], gold[f"item{k}_pose2d"])
        assert np.array_equal(table.features[idx][:, ::64], gold[f"item{k}_img_feature_sub"])
        assert np.array_equal(gt[idx[8]], gold[f"item{k}_reg_pose3d"])                                          # the middle frame's target
    seq = table.sequence_ids()
    assert seq[0] == 0 and len(np.unique(seq)) == 4 and np.all(np.diff(seq) >= 0)


def test_missing_files_are_named(tmp_path):
    from pmce_amd import datasets
    with pytest.raises(FileNotFoundError, match="3DPW_latest_test.json"):
        datasets.load_pw3d(str(tmp_path))
