#!/usr/bin/env python3
"""Golden vectors for pmce_amd/datasets.py from the REAL reference loader (build container only): the reference's own ``PW3D`` dataset
class (data/PW3D/dataset.py:14-258) is instantiated on the small synthetic 3DPW-format directory that tests/golden/pw3d_files.py writes,
and its results are stored: the sorted frame list, the per-frame arrays ``load_data`` returns, the window list ``vid_indices`` and, for a
few windows, what ``__getitem__`` hands the model (``pose2d``, ``img_feature``) and the joint target.

Stand-ins for what the image lacks (none of them reference code, none restates arithmetic of the path): pycocotools' COCO (a JSON index:
``anns``, ``loadImgs``), the SMPL layer (zeros: the mesh target is not stored), core.config, and skimage's view_as_windows
(numpy's sliding_window_view), as in make_golden_staging.py."""
import json
import os
import os.path as osp
import sys
import tempfile
import types

import numpy as np
import torch

HERE = osp.dirname(osp.abspath(__file__)); REPO = osp.dirname(osp.dirname(HERE)); REF = "/root/reference"
sys.path.insert(0, REPO); sys.path.insert(0, HERE)
import pw3d_files  # noqa: E402


class _COCO:
    """the part of pycocotools.coco.COCO the loader uses: annotations in file order, images by id"""
    def __init__(self, path=None):
        d = json.load(open(path))
        self.anns = {a["id"]: a for a in d["annotations"]}
        self.imgs = {i["id"]: i for i in d["images"]}

    def loadImgs(self, ids):
        return [self.imgs[i] for i in (ids if isinstance(ids, (list, tuple)) else [ids])]


class _SMPL:
    root_joint_idx, face_kps_vertex = 0, (0,)
    joint_regressor_h36m = np.zeros((17, 6890), np.float32)
    joint_regressor_coco = np.zeros((19, 6890), np.float32)

    def __init__(self):
        lay = lambda pose, shape, trans: (torch.zeros(1, 6890, 3), torch.zeros(1, 24, 3))
        neutral = types.SimpleNamespace(th_J_regressor=torch.zeros(24, 6890), __call__=None)
        self.layer = {"neutral": neutral, "male": lay, "female": lay}


def shims(data_dir):
    class AD(dict):
        __getattr__ = dict.__getitem__
    core = types.ModuleType("core"); cc = types.ModuleType("core.config")
    cc.cfg = AD(data_dir=data_dir, TEST=AD(vis=False), vis_dir="/tmp", DATASET=AD(seqlen=16, stride=1, use_gt_input=False), MODEL=AD(name="PMCE"))
    core.config = cc
    pc = types.ModuleType("pycocotools"); pcc = types.ModuleType("pycocotools.coco"); pcc.COCO = _COCO; pc.coco = pcc
    fu = types.ModuleType("funcs_utils"); fu.save_obj = lambda *a, **k: None
    sm = types.ModuleType("smpl"); sm.SMPL = _SMPL
    sk = types.ModuleType("skimage"); sku = types.ModuleType("skimage.util"); skus = types.ModuleType("skimage.util.shape")
    skus.view_as_windows = lambda arr, window_shape, step=1: np.lib.stride_tricks.sliding_window_view(arr, window_shape)[::step]
    sku.shape = skus; sk.util = sku
    cv2 = types.ModuleType("cv2")
    tv = types.ModuleType("torchvision"); tvt = types.ModuleType("torchvision.transforms"); tv.transforms = tvt
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt, "core": core, "core.config": cc, "pycocotools": pc,
                        "pycocotools.coco": pcc, "funcs_utils": fu, "smpl": sm, "skimage": sk, "skimage.util": sku,
                        "skimage.util.shape": skus, "cv2": cv2})
    sys.path.insert(0, osp.join(REF, "lib")); sys.path.insert(0, osp.join(REF, "data"))


SAMPLE_WINDOWS = (0, 5, 21, 22, 40, 66)


def main():
    root = tempfile.mkdtemp()
    pw3d_files.write(root)
    shims(root)
    from PW3D.dataset import PW3D
    ds = PW3D("test", None)
    out = {"img_paths": np.array(ds.img_paths), "vid_names": np.array(ds.vid_names), "img_shapes": ds.img_shapes,
           "pred_pose2ds": ds.pred_pose2ds, "features_sub": ds.features.astype(np.float32)[:, ::64],     # every 64th channel (the file's values, copied: a subsample pins the row order)
           "joints_cam_h36m": ds.joints_cam_h36m,
           "joints_cam_coco": ds.joints_cam_coco, "vid_indices": np.asarray(ds.vid_indices).reshape(-1, 2), "n_items": np.int64(len(ds)),
           "sample_windows": np.array(SAMPLE_WINDOWS)}
    for k in SAMPLE_WINDOWS:
        inputs, targets, meta = ds[k]
        out[f"item{k}_pose2d"] = np.asarray(inputs["pose2d"], dtype=np.float32)
        out[f"item{k}_img_feature_sub"] = np.asarray(inputs["img_feature"], dtype=np.float32)[:, ::64]
        out[f"item{k}_reg_pose3d"] = np.asarray(targets["reg_pose3d"], dtype=np.float32)
    np.savez_compressed(osp.join(HERE, "datasets_pw3d.npz"), **out)
    print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if not k.startswith("item")}, len(ds))


if __name__ == "__main__":
    main()
