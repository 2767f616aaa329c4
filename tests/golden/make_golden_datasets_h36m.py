#!/usr/bin/env python3
"""Golden vectors for pmce_amd.datasets.load_h36m from the REAL reference loader (build container only): the reference's own ``Human36M``
dataset class (data/Human36M/dataset.py:25-560, test split, protocol 2, the 'human36' input joint set of config/test_mesh_h36m.yml) is
instantiated on the small synthetic Human3.6M-format directory that tests/golden/h36m_files.py writes, and its results are stored: the frame
list, the per-frame arrays ``load_data`` / ``load_pose2d_det`` return, the window list ``vid_indices`` and, for a few windows, what
``__getitem__`` hands the model (``pose2d``, ``img_feature``) and the joint target.

Stand-ins for what the image lacks (none of them reference code, none restates arithmetic of the path): pycocotools' COCO (a JSON index:
``dataset``, ``createIndex``, ``anns``, ``loadImgs``), the SMPL layer and ``Human36M.get_smpl_coord`` (zeros: SMPL model files are out of
scope and the mesh target is not stored), core.config, transforms3d / cv2 / easydict / matplotlib (imported, not used on this path), numpy < 1.24's
object arrays for ragged rows (the installed numpy raises instead) and
skimage's view_as_windows (numpy's sliding_window_view), as in make_golden_datasets.py."""
import os.path as osp
import sys
import tempfile
import types

import numpy as np
import torch

HERE = osp.dirname(osp.abspath(__file__)); REPO = osp.dirname(osp.dirname(HERE)); REF = "/root/reference"
sys.path.insert(0, REPO); sys.path.insert(0, HERE)
import h36m_files  # noqa: E402


class _COCO:
    """the part of pycocotools.coco.COCO the loader uses: an empty index filled through ``dataset``, then ``createIndex``"""
    def __init__(self, path=None):
        self.dataset, self.anns, self.imgs = {}, {}, {}

    def createIndex(self):
        self.anns = {a["id"]: a for a in self.dataset.get("annotations", [])}
        self.imgs = {i["id"]: i for i in self.dataset.get("images", [])}

    def loadImgs(self, ids):
        return [self.imgs[i] for i in (ids if isinstance(ids, (list, tuple)) else [ids])]


class _SMPL:
    root_joint_idx, face_kps_vertex = 0, (0,)
    joint_regressor_h36m = np.zeros((17, 6890), np.float32)
    joint_regressor_coco = np.zeros((19, 6890), np.float32)

    def __init__(self):
        self.layer = {"neutral": types.SimpleNamespace(th_J_regressor=torch.zeros(24, 6890))}


def shims(data_dir):
    class AD(dict):
        __getattr__ = dict.__getitem__
    core = types.ModuleType("core"); cc = types.ModuleType("core.config")
    cc.cfg = AD(data_dir=data_dir, TEST=AD(vis=False), vis_dir="/tmp", DATASET=AD(seqlen=16, stride=1, use_gt_input=False, input_joint_set="human36"),
                MODEL=AD(name="PMCE", input_shape=(384, 288)))
    core.config = cc
    pc = types.ModuleType("pycocotools"); pcc = types.ModuleType("pycocotools.coco"); pcc.COCO = _COCO; pc.coco = pcc
    fu = types.ModuleType("funcs_utils"); fu.save_obj = lambda *a, **k: None
    sm = types.ModuleType("smpl"); sm.SMPL = _SMPL
    sk = types.ModuleType("skimage"); sku = types.ModuleType("skimage.util"); skus = types.ModuleType("skimage.util.shape")
    skus.view_as_windows = lambda arr, window_shape, step=1: np.lib.stride_tricks.sliding_window_view(arr, window_shape)[::step]
    sku.shape = skus; sk.util = sku
    mods = {"core": core, "core.config": cc, "pycocotools": pc, "pycocotools.coco": pcc, "funcs_utils": fu, "smpl": sm, "skimage": sk,
            "skimage.util": sku, "skimage.util.shape": skus}
    for name in ("cv2", "transforms3d", "torchvision", "torchvision.transforms", "matplotlib", "matplotlib.pyplot"):
        mods[name] = types.ModuleType(name)
    mods["torchvision"].transforms = mods["torchvision.transforms"]
    mods["matplotlib"].pyplot = mods["matplotlib.pyplot"]
    ed = types.ModuleType("easydict"); ed.EasyDict = AD; mods["easydict"] = ed
    sys.modules.update(mods)
    sys.path.insert(0, osp.join(REF, "lib")); sys.path.insert(0, osp.join(REF, "data"))


SAMPLE_WINDOWS = (0, 3, 17, 18, 40, -1)


def main():
    root = tempfile.mkdtemp()
    h36m_files.write(root)
    shims(root)
    from Human36M.dataset import Human36M
    ref_mod = sys.modules["Human36M.dataset"]      # (the package rebinds the name `dataset` to the class)

    class _OldNumpy:      # the reference relies on numpy < 1.24: np.array of ragged rows (frames without an SMPL fit) is an OBJECT array there
        def __getattr__(self, name):
            return getattr(np, name)

        @staticmethod
        def array(obj, *a, **k):
            try:
                return np.array(obj, *a, **k)
            except ValueError:
                out = np.empty(len(obj), dtype=object)
                for i, o in enumerate(obj):
                    out[i] = o
                return out
    ref_mod.np = _OldNumpy()
    Human36M.get_smpl_coord = lambda self, smpl_param, cam_param: (np.zeros((6890, 3), np.float32), np.zeros((24, 3), np.float32))
    ds = Human36M("test", types.SimpleNamespace(debug=False))
    valid = np.array([len(p) != 1 for p in ds.poses])
    out = {"img_names": np.array(ds.img_names), "img_hws": ds.img_hws, "bboxs": ds.bboxs.astype(np.float32), "joint_imgs": ds.joint_imgs,
           "joint_cams": ds.joint_cams, "cam_idxs": np.asarray(ds.cam_idxs), "smpl_valid": valid,
           "poses_valid": np.stack([p for p in ds.poses if len(p) != 1]).astype(np.float32),
           "features_sub": ds.features.astype(np.float32)[:, ::64],          # every 64th channel (the file's values, copied: a subsample pins the row order)
           "cam_focals": ds.cam_param_focals, "cam_princpts": ds.cam_param_princpts, "cam_Rs": ds.cam_param_Rs, "cam_ts": ds.cam_param_ts,
           "pose2d_det": ds.datalist_pose2d_det, "pose2d_det_name": np.array(ds.datalist_pose2d_det_name),
           "vid_indices": np.asarray(ds.vid_indices).reshape(-1, 2), "n_items": np.int64(len(ds)), "sample_windows": np.array(SAMPLE_WINDOWS)}
    for k in SAMPLE_WINDOWS:
        inputs, targets, meta = ds[k]
        out[f"item{k}_pose2d"] = np.asarray(inputs["pose2d"], dtype=np.float32)
        out[f"item{k}_img_feature_sub"] = np.asarray(inputs["img_feature"], dtype=np.float32)[:, ::64]
        out[f"item{k}_reg_pose3d"] = np.asarray(targets["reg_pose3d"], dtype=np.float32)
        out[f"item{k}_lift_pose3d"] = np.asarray(targets["lift_pose3d"], dtype=np.float32)
    np.savez_compressed(osp.join(HERE, "datasets_h36m.npz"), **out)
    print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if not k.startswith("item")}, len(ds))


if __name__ == "__main__":
    main()
