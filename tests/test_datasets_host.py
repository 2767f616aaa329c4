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


# ---- Human3.6M (round 5): the reference's Human36M('test') class on tests/golden/h36m_files.py's directory (make_golden_datasets_h36m.py) ----

@pytest.fixture(scope="module")
def h36m(tmp_path_factory):
    import h36m_files
    from pmce_amd import datasets
    root = str(tmp_path_factory.mktemp("h36m"))
    return datasets.load_h36m(h36m_files.write(root), "test")


@pytest.fixture(scope="module")
def gold_h36m():
    return np.load(osp.join(HERE, "golden", "datasets_h36m.npz"))


def test_h36m_frame_tables_equal_the_reference_loader(h36m, gold_h36m):
    """Every second frame, the sequence dropped by name and the empty bounding box gone, features walked from each video's start index,
    joints projected world -> camera -> pixel, detections in the dataset's order: every array the reference's load_data returns."""
    g = gold_h36m
    assert len(h36m) == len(g["img_names"]) == 155 and h36m.skipped == 18 + 1
    assert list(h36m.img_paths) == list(g["img_names"])
    assert np.array_equal(h36m.img_shapes, g["img_hws"])
    assert np.array_equal(h36m.features[:, ::64], g["features_sub"])
    assert np.array_equal(h36m.joints_cam_h36m, g["joint_cams"]) and np.array_equal(h36m.gt_joints_img_coco, g["joint_imgs"])
    assert np.array_equal(h36m.extras["bboxs"], g["bboxs"])
    assert np.array_equal(h36m.cam_idxs, g["cam_idxs"]) and np.array_equal(h36m.mid_valid, g["smpl_valid"])
    assert np.array_equal(h36m.smpl["pose"][h36m.mid_valid], g["poses_valid"])
    for k, name in (("cam_focals", "cam_focals"), ("cam_princpts", "cam_princpts"), ("cam_Rs", "cam_Rs"), ("cam_ts", "cam_ts")):
        assert np.array_equal(h36m.extras[k], g[name]), k
    assert np.array_equal(h36m.keypoints, g["pose2d_det"]) and list(h36m.img_paths) == list(g["pose2d_det_name"])


def test_h36m_window_list_and_model_inputs_equal_the_reference_dataset(h36m, gold_h36m):
    """vid_indices (windows whose middle frame has no SMPL fit dropped) and, for sampled windows, what __getitem__ hands the model and the
    joint targets of the middle frame."""
    from oracle import staging_oracle as S
    from pmce_amd import datasets
    g = gold_h36m
    win = h36m.windows(16, 1)
    assert np.array_equal(win, g["vid_indices"]) and len(win) == int(g["n_items"]) == 76
    pose2d = np.stack([np.asarray(S.normalize_screen_coordinates(h36m.keypoints[i][:, :2], w=h36m.img_shapes[i][1], h=h36m.img_shapes[i][0]), dtype=np.float32)
                       for i in range(len(h36m))])
    gt = h36m.gt_joints_root_relative()
    frames = datasets.window_frames(win)
    for k in g["sample_windows"]:
        idx = frames[k]
        assert np.array_equal(pose2d[idx], g[f"item{k}_pose2d"])
        assert np.array_equal(h36m.features[idx][:, ::64], g[f"item{k}_img_feature_sub"])
        assert np.array_equal(gt[idx[8]], g[f"item{k}_reg_pose3d"]) and np.array_equal(gt[idx[8]], g[f"item{k}_lift_pose3d"])
    assert h36m.extra_joints == 0 and len(h36m.joints_name) == 17
    cam4 = h36m.cam_idxs[frames[:, 8]] == 4                         # what Human36M.evaluate keeps
    assert 0 < cam4.sum() < len(win)


def test_h36m_inconsistent_files_are_refused(tmp_path):
    import json
    import h36m_files
    from pmce_amd import datasets
    path = h36m_files.write(str(tmp_path))
    with pytest.raises(FileNotFoundError, match="Human36M_subject9_data.json"):
        datasets.load_h36m(str(tmp_path / "nowhere"))
    f = osp.join(path, "Human36M_test_cpn_joint_2d.json")
    det = json.load(open(f))
    det.pop(sorted(det)[0])
    json.dump(det, open(f, "w"))
    with pytest.raises(ValueError, match="no CPN detection"):
        datasets.load_h36m(path)


# ---- MPI-INF-3DHP (config/test_mesh_mpii3d.yml) ---------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def mpii3d_table(tmp_path_factory):
    import mpii3d_files
    from pmce_amd import datasets
    root = str(tmp_path_factory.mktemp("mpii3d"))
    return datasets.load_mpii3d(mpii3d_files.write(root), "val")


def test_mpii3d_tables_windows_and_model_inputs_equal_the_reference_loader(mpii3d_table):
    """pmce_amd.datasets.load_mpii3d against the reference's own ``MPII3D('test')`` (data/MPII3D/dataset.py:249-292,467-517) on the synthetic
    directory of tests/golden/mpii3d_files.py (fixture: make_golden_datasets_mpii3d.py): rows sorted by image name, the 49 SPIN joints as
    Human3.6M joints in millimetres bit for bit (convert_kps + transform_joint_to_other_db), 2048 x 2048 images, ``vid_indices``
    (split_into_chunks_pose with the VIBE tail rule; the 11-frame video gives no window) and, for sampled windows, what ``__getitem__`` hands
    the model and the joint target."""
    from oracle import staging_oracle as S
    from pmce_amd import datasets
    t = mpii3d_table
    g = np.load(osp.join(HERE, "golden", "datasets_mpii3d.npz"))
    assert len(t) == 101 and list(t.img_paths) == list(g["img_paths"]) and np.array_equal(t.img_shapes, g["img_shapes"])
    assert np.array_equal(t.features[:, ::64], g["features_sub"]) and np.array_equal(t.joints_cam_h36m, g["joints_cam"])
    ext = np.stack([S.add_pelvis_and_neck(k) for k in t.keypoints])
    assert np.array_equal(ext, g["pred_pose2ds"])
    win = t.windows(16, 1)
    assert np.array_equal(win, g["vid_indices"]) and len(win) == int(g["n_items"]) == 50
    pose2d = np.stack([np.asarray(S.normalize_screen_coordinates(ext[i][:, :2], w=2048, h=2048), dtype=np.float32) for i in range(len(t))])
    gt = t.gt_joints_root_relative()
    frames = datasets.window_frames(win)
    for k in g["sample_windows"]:
        idx = frames[k]
        assert np.array_equal(pose2d[idx], g[f"item{k}_pose2d"])
        assert np.array_equal(t.features[idx][:, ::64], g[f"item{k}_img_feature_sub"])
        assert np.array_equal(gt[idx[8]], g[f"item{k}_reg_pose3d"])
    seq = t.sequence_ids()
    assert len(np.unique(seq)) == 3 and np.all(np.diff(seq) >= 0)


def test_mpii3d_missing_files_and_detections_are_named(tmp_path):
    import json
    import mpii3d_files
    from pmce_amd import datasets
    with pytest.raises(FileNotFoundError, match="mpii3d_val_scale12_db.pt"):
        datasets.load_mpii3d(str(tmp_path))
    path = mpii3d_files.write(str(tmp_path))
    vit = json.load(open(osp.join(path, "vitpose_mpii3d_val_output.json")))
    json.dump(vit[1:], open(osp.join(path, "vitpose_mpii3d_val_output.json"), "w"))
    with pytest.raises(ValueError, match="no detection for 1 image"):
        datasets.load_mpii3d(path)
