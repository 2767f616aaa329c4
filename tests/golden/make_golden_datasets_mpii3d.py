#!/usr/bin/env python3
"""Golden vectors for pmce_amd.datasets.load_mpii3d from the REAL reference loader (build container only): the reference's own ``MPII3D``
dataset class (data/MPII3D/dataset.py:21-105,249-292,467-517) is instantiated on the small synthetic directory that
tests/golden/mpii3d_files.py writes, and its results are stored: the sorted frame list, the per-frame arrays ``load_data_val`` returns
(joints through ``convert_kps`` and ``transform_joint_to_other_db``, millimetres), the window list ``vid_indices``
(``split_into_chunks_pose``) and, for a few windows, what ``__getitem__`` hands the model and the joint target.

Stand-ins for what the image lacks: those of make_golden_datasets_h36m.py (pycocotools, SMPL layer, core.config, skimage, cv2, torchvision,
transforms3d, easydict, matplotlib: imported by the dataset modules, not used on this path)."""
import os.path as osp
import sys
import tempfile
import types

import numpy as np

HERE = osp.dirname(osp.abspath(__file__)); REPO = osp.dirname(osp.dirname(HERE))
sys.path.insert(0, REPO); sys.path.insert(0, HERE)
import make_golden_datasets_h36m as G  # noqa: E402
import mpii3d_files  # noqa: E402

SAMPLE_WINDOWS = (0, 7, 32, 33, 49)


def main():
    root = tempfile.mkdtemp()
    mpii3d_files.write(root)
    G.shims(root)
    sys.modules["core.config"].cfg["DATASET"]["input_joint_set"] = "coco"
    from MPII3D.dataset import MPII3D
    ds = MPII3D("test", types.SimpleNamespace(debug=False))
    out = {"img_paths": np.array(ds.img_paths), "img_shapes": ds.img_shapes, "pred_pose2ds": ds.pred_pose2ds,
           "features_sub": ds.img_feats.astype(np.float32)[:, ::64], "joints_cam": ds.joints_cam,
           "vid_indices": np.asarray(ds.vid_indices).reshape(-1, 2), "n_items": np.int64(len(ds.vid_indices)),
           "sample_windows": np.array(SAMPLE_WINDOWS)}
    for k in SAMPLE_WINDOWS:
        inputs, targets, meta = ds[k]
        out[f"item{k}_pose2d"] = np.asarray(inputs["pose2d"], dtype=np.float32)
        out[f"item{k}_img_feature_sub"] = np.asarray(inputs["img_feature"], dtype=np.float32)[:, ::64]
        out[f"item{k}_reg_pose3d"] = np.asarray(targets["reg_pose3d"], dtype=np.float32)
    np.savez_compressed(osp.join(HERE, "datasets_mpii3d.npz"), **out)
    print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if not k.startswith("item")})


if __name__ == "__main__":
    main()
