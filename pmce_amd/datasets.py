"""Readers for the reference's precomputed-feature / detection files (BASELINE configs 3-4): what
``data/PW3D/dataset.py:90-183`` (``PW3D.load_data``) reads, parsed into PER-FRAME tables that feed the GPU-side staging of this
package - ``staging.prepare_pose2d`` (pelvis / neck + screen normalisation, dataset.py:185-204 + :241-243), ``staging.mesh_window_table``
(``split_into_chunks_mesh``, lib/_img_utils.py:58-92) and the on-device window assembly (``streaming``).  Host Python, no pycocotools:
the annotation file is plain COCO-style JSON.

    table = datasets.load_pw3d("/data/PW3D/pw3d_data")            # or load_pw3d(cfg.data_dir + "/PW3D/pw3d_data", "test")
    win = table.windows()                                         # [W, 2] inclusive (start, end) frame indices, reference order
    pose2d = table.pose2d(device)                                 # [N, 19, 2] normalised screen coordinates, on the GPU
    feats = table.features_on(device)                             # [N, 2048]
    p, f = datasets.window_batch(pose2d, feats, win[a:b])         # model inputs [b-a, 16, 19, 2], [b-a, 16, 2048]

What is NOT here: the SMPL layer that turns ``smpl_param`` into ground-truth meshes (``PW3D.get_smpl_coord``; SMPL model files are
out of scope) - ground-truth joints come from ``3DPW_<split>_joint_h36m_cam.json`` as the reference's ``reg_pose3d`` target
(dataset.py:229-230,249), and a ground-truth mesh table can be supplied as ``<data_path>/3DPW_<split>_gt_mesh_cam.npy`` ([N, 6890, 3]
float32 mm, root-relative, in the table's frame order) by whoever holds the SMPL files.  The Human3.6M loader
(data/Human36M/dataset.py:194-350: per-subject annotation / camera / SMPL-fit files, bounding-box processing) is not restated.
"""
from __future__ import annotations

import json
import os.path as osp
from dataclasses import dataclass, field

import numpy as np

COCO_JOINTS = ('Nose', 'L_Eye', 'R_Eye', 'L_Ear', 'R_Ear', 'L_Shoulder', 'R_Shoulder', 'L_Elbow', 'R_Elbow', 'L_Wrist',
               'R_Wrist', 'L_Hip', 'R_Hip', 'L_Knee', 'R_Knee', 'L_Ankle', 'R_Ankle')
PW3D_FILES = ("3DPW_latest_{s}.json", "vitpose_3dpw_{s}_output.json", "3DPW_{s}_joint_coco_cam.json",
              "3DPW_{s}_gt_joint_coco_img.json", "3DPW_{s}_joint_h36m_cam.json", "3DPW_{s}_img_feat.json")


@dataclass
class FrameTable:
    """Per-frame tables of one dataset split, sorted by image path (person/sequence/image_xxxxx.jpg) as the reference sorts them
    (dataset.py:175-181): consecutive rows of one (sequence, person) are consecutive frames of one video."""
    name: str
    img_paths: np.ndarray            # str [N]
    vid_names: np.ndarray            # str [N]  sequence + person id (dataset.py:137)
    img_shapes: np.ndarray           # int32 [N, 2]  (height, width)
    keypoints: np.ndarray            # float32 [N, 17, 3]  detector output in pixels: x, y, score (COCO order)
    features: np.ndarray             # float32 [N, 2048]
    joints_cam_h36m: np.ndarray      # float32 [N, 17, 3]  mm, camera coordinates
    joints_cam_coco: np.ndarray      # float32 [N, 19, 3]
    gt_joints_img_coco: np.ndarray   # float32 [N, 17|19, 2|3]
    smpl: dict = field(default_factory=dict)     # 'pose' [N,72], 'shape' [N,10], 'trans' [N,3], 'gender' str [N]
    gt_mesh_cam: np.ndarray = None   # optional float32 [N, 6890, 3] mm, root-relative (see the module docstring)
    skipped: int = 0                 # annotations without a feature entry (dataset.py:148-151)

    def __len__(self):
        return len(self.img_paths)

    def windows(self, seqlen: int = 16, stride: int = 1) -> np.ndarray:
        """``self.vid_indices`` of the reference dataset (dataset.py:62): every 3DPW frame carries a full SMPL pose, so no window is
        dropped for an invalid middle frame."""
        from .staging import mesh_window_table
        return mesh_window_table(list(self.img_paths), seqlen, stride, None)

    def sequence_ids(self) -> np.ndarray:
        """int id of every frame's video (first appearance order): the grouping key of the acceleration error."""
        _, first, inv = np.unique(self.vid_names, return_index=True, return_inverse=True)
        rank = np.empty(len(first), dtype=np.int64)
        rank[np.argsort(first)] = np.arange(len(first))
        return rank[inv]

    def pose2d(self, device):
        """[N, 19, 2] model input coordinates on `device`: pelvis / neck appended, ``X / w * 2 - [1, h / w]`` (one kernel)."""
        import torch
        from .staging import prepare_pose2d
        return prepare_pose2d(torch.from_numpy(self.keypoints).to(device), torch.from_numpy(self.img_shapes).to(device), COCO_JOINTS, 2)

    def features_on(self, device):
        import torch
        return torch.from_numpy(self.features).to(device)

    def gt_joints_root_relative(self) -> np.ndarray:
        """The reference's ``reg_pose3d`` target: h36m joints minus their root (dataset.py:229-230), mm."""
        return self.joints_cam_h36m - self.joints_cam_h36m[:, :1]


def load_pw3d(data_path: str, split: str = "test") -> FrameTable:
    """Parse the six files ``PW3D.load_data`` reads (test / validation splits: detector keypoints from the ViTPose output file)."""
    need = [osp.join(data_path, f.format(s=split)) for f in PW3D_FILES]
    missing = [p for p in need if not osp.exists(p)]
    if missing:
        raise FileNotFoundError("3DPW files missing under %s: %s" % (data_path, ", ".join(osp.basename(p) for p in missing)))
    load = lambda p: json.load(open(p))
    db, vit, coco_cam, gt_img, h36m_cam, raw_feats = (load(p) for p in need)
    images = {im["id"]: im for im in db["images"]}
    det = {str(item["annotation_id"]): np.asarray(item["keypoints"], dtype=np.float32)[:, :3] for item in vit}
    rows, skipped = [], 0
    for ann in db["annotations"]:                                 # pycocotools keeps the file's order (dict of ann id)
        img = images[ann["image_id"]]
        seq, name, pid = img["sequence"], img["file_name"], ann["person_id"]
        s, i, p = str(seq), str(int(name[6:-4])), str(int(pid))
        feat = raw_feats.get(f"{s}_{p}_{i}")
        if feat is None:
            skipped += 1
            continue
        sp = ann["smpl_param"]
        rows.append((osp.join(str(pid), seq, name), seq + str(pid), (img["height"], img["width"]), det[str(int(ann["id"]))],
                     feat, h36m_cam[s][i][p], coco_cam[s][i][p], gt_img[s][i][p], sp["pose"], sp["shape"], sp["trans"], sp["gender"]))
    cols = list(zip(*rows))
    img_paths = np.array(cols[0])
    perm = np.argsort(img_paths)
    take = lambda c, dt: np.asarray(c, dtype=dt)[perm]
    table = FrameTable(
        name=f"3DPW {split}", img_paths=img_paths[perm], vid_names=np.array(cols[1])[perm], img_shapes=take(cols[2], np.int32),
        keypoints=take(cols[3], np.float32), features=take(cols[4], np.float32), joints_cam_h36m=take(cols[5], np.float32),
        joints_cam_coco=take(cols[6], np.float32), gt_joints_img_coco=take(cols[7], np.float32),
        smpl={"pose": take(cols[8], np.float32), "shape": take(cols[9], np.float32), "trans": take(cols[10], np.float32),
              "gender": np.array(cols[11])[perm]}, skipped=skipped)
    mesh_file = osp.join(data_path, f"3DPW_{split}_gt_mesh_cam.npy")
    if osp.exists(mesh_file):
        m = np.load(mesh_file, mmap_mode="r")
        if m.shape != (len(table), 6890, 3):
            raise ValueError(f"{mesh_file}: expected shape {(len(table), 6890, 3)} (the table's frame order), got {m.shape}")
        table.gt_mesh_cam = m
    return table


def window_frames(win: np.ndarray, seqlen: int = 16) -> np.ndarray:
    """[W, seqlen] frame indices of windows given as inclusive (start, end) pairs; start == end repeats one frame (dataset.py:213-216)."""
    win = np.asarray(win).reshape(-1, 2)
    step = (win[:, 1] != win[:, 0]).astype(np.int64)
    return win[:, :1] + step[:, None] * np.arange(seqlen)[None, :]


def window_batch(pose2d_frames, feat_frames, win, seqlen: int = 16):
    """Model inputs of a batch of windows, gathered ON THE DEVICE from the per-frame tables (each 8 KB frame feature was uploaded
    once, not 16 times): (pose2d [W, seqlen, J, 2], img_feat [W, seqlen, 2048])."""
    import torch
    idx = torch.from_numpy(window_frames(win, seqlen)).to(pose2d_frames.device)
    return pose2d_frames[idx], feat_frames[idx]
