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
float32 mm, root-relative, in the table's frame order) by whoever holds the SMPL files.

``load_h36m`` does the same for ``data/Human36M/dataset.py:105-130,194-350`` (``Human36M.load_data`` + ``load_pose2d_det``; test split,
the 'human36' input joint set of config/test_mesh_h36m.yml): per-subject annotation / camera / world-joint / SMPL-fit files, the joblib
feature database walked with each video's start index, the CPN detections, every second frame, the reference's drops (one sequence by name,
empty bounding boxes), world -> camera -> pixel projection of the annotated joints.  The 'coco' joint set of the training configs (NeuralAnnot
COCO joints + noise files) is not restated.

``load_mpii3d`` reads the MPI-INF-3DHP validation files of ``data/MPII3D/dataset.py:249-292`` (config/test_mesh_mpii3d.yml): the joblib
database and the ViTPose detections; the training split (NeuralAnnot SMPL fits, camera files) is not restated.
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
    skipped: int = 0                 # annotations without a feature entry (dataset.py:148-151) / dropped by the Human3.6M rules
    joints_name: tuple = None        # names of the keypoints' joints (default: COCO-17) and how many joints the model input appends to them
    extra_joints: int = 2            # (2: pelvis + neck, the COCO-19 input; 0: the Human3.6M 17-joint input)
    mid_valid: np.ndarray = None     # bool [N]: the frame has an SMPL fit - a window whose middle frame has none is dropped (lib/_img_utils.py:75)
    cam_idxs: np.ndarray = None      # int [N]: Human3.6M camera of the frame (the evaluation keeps camera 4, Human36M/dataset.py:742-744)
    extras: dict = field(default_factory=dict)   # whatever else the reference's load_data returns (bboxs, joint_imgs, camera parameters)

    def __len__(self):
        return len(self.img_paths)

    def windows(self, seqlen: int = 16, stride: int = 1) -> np.ndarray:
        """``self.vid_indices`` of the reference dataset (PW3D/dataset.py:62, Human36M/dataset.py:98-101): every 3DPW frame carries a full
        SMPL pose; a Human3.6M window whose middle frame has no SMPL fit is dropped."""
        from .staging import mesh_window_table
        return mesh_window_table(list(self.img_paths), seqlen, stride, self.mid_valid)

    def pose_windows(self, seqlen: int = 16, stride: int = 1) -> np.ndarray:
        """``self.vid_indices`` of the reference's POSE-ONLY test configurations (cfg.MODEL.name == 'PoseEst'): Human3.6M uses
        ``split_into_chunks_pose`` - no SMPL-fit filter (Human36M/dataset.py:99-100); 3DPW uses the mesh table there too (PW3D/dataset.py:61)."""
        from .staging import pose_window_table
        if self.cam_idxs is None:
            return self.windows(seqlen, stride)
        return pose_window_table(list(self.img_paths), seqlen, stride)

    def sequence_ids(self) -> np.ndarray:
        """int id of every frame's video (first appearance order): the grouping key of the acceleration error."""
        _, first, inv = np.unique(self.vid_names, return_index=True, return_inverse=True)
        rank = np.empty(len(first), dtype=np.int64)
        rank[np.argsort(first)] = np.arange(len(first))
        return rank[inv]

    def pose2d(self, device):
        """[N, J, 2] model input coordinates on `device`: pelvis / neck appended for the COCO-19 input, ``X / w * 2 - [1, h / w]`` (one kernel)."""
        import torch
        from .staging import prepare_pose2d
        return prepare_pose2d(torch.from_numpy(np.ascontiguousarray(self.keypoints)).to(device), torch.from_numpy(self.img_shapes).to(device),
                              self.joints_name or COCO_JOINTS, self.extra_joints)

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
        # the three joint tables are indexed BEFORE the feature lookup, as the reference does (PW3D/dataset.py:144-151): a frame missing
        # from them raises (KeyError, as there) even when it has no feature either; only a missing FEATURE skips the annotation
        jc, ji, jh = coco_cam[s][i][p], gt_img[s][i][p], h36m_cam[s][i][p]
        feat = raw_feats.get(f"{s}_{p}_{i}")
        if feat is None:
            skipped += 1
            continue
        aid = str(int(ann["id"]))
        if aid not in det:
            raise ValueError(f"vitpose_3dpw_{split}_output.json holds no detection for annotation id {aid} ({seq}/{name}, person {pid})")
        sp = ann["smpl_param"]
        rows.append((osp.join(str(pid), seq, name), seq + str(pid), (img["height"], img["width"]), det[aid],
                     feat, jh, jc, ji, sp["pose"], sp["shape"], sp["trans"], sp["gender"]))
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


H36M_JOINTS = ('Pelvis', 'R_Hip', 'R_Knee', 'R_Ankle', 'L_Hip', 'L_Knee', 'L_Ankle', 'Torso', 'Neck', 'Nose', 'Head',
               'L_Shoulder', 'L_Elbow', 'L_Wrist', 'R_Shoulder', 'R_Elbow', 'R_Wrist')
H36M_TEST_SUBJECTS = {1: (11,), 2: (9, 11)}          # protocol -> subjects (Human36M/dataset.py:184-188)
H36M_DROPPED_SEQUENCE = "s_11_act_02_subact_02_ca_0"  # (dataset.py:255-256)


def _sanitized_bbox(bbox, aspect_ratio):
    """``process_bbox`` (lib/coord_utils.py:66-90): None for an empty box, else the aspect-ratio preserving box (float64, as there)."""
    x, y, w, h = bbox
    x1, y1, x2, y2 = x, y, x + (w - 1), y + (h - 1)
    if not (w * h > 0 and x2 >= x1 and y2 >= y1):
        return None
    b = np.array([x1, y1, x2 - x1, y2 - y1])
    w, h = b[2], b[3]
    cx, cy = b[0] + w / 2., b[1] + h / 2.
    if w > aspect_ratio * h:
        h = w / aspect_ratio
    elif w < aspect_ratio * h:
        w = h * aspect_ratio
    b[2], b[3] = w, h
    b[0], b[1] = cx - b[2] / 2., cy - b[3] / 2.
    return b


def load_h36m(data_path: str, split: str = "test", protocol: int = 2, sampling_ratio: int = 2, aspect_ratio: float = 288 / 384) -> FrameTable:
    """Parse what ``Human36M('test')`` reads with ``input_joint_set = 'human36'`` (Human36M/dataset.py:105-130,194-350) into per-frame tables
    in the reference's frame order (annotation order of subject 9's file, then subject 11's).  ``aspect_ratio`` = cfg.MODEL.input_shape[1] /
    input_shape[0] (core/config.py:63; it only shapes the stored boxes - an EMPTY box drops the frame whatever the ratio)."""
    import joblib
    if split != "test":
        raise ValueError("load_h36m restates the test split (detections from Human36M_test_cpn_joint_2d.json)")
    annot = osp.join(data_path, "annotations")
    subjects = H36M_TEST_SUBJECTS[protocol]
    need = [osp.join(data_path, f"h36m_{split}_imgfeat_db_concat.pt"), osp.join(data_path, f"Human36M_{split}_start_idx_tight.json"),
            osp.join(data_path, f"Human36M_{split}_cpn_joint_2d.json")]
    need += [osp.join(annot, f"Human36M_subject{s}_{kind}.json") for s in subjects for kind in ("data", "camera", "joint_3d", "SMPL_NeuralAnnot")]
    missing = [p for p in need if not osp.exists(p)]
    if missing:
        raise FileNotFoundError("Human3.6M files missing under %s: %s" % (data_path, ", ".join(osp.basename(p) for p in missing)))
    load = lambda p: json.load(open(p))
    img_db = joblib.load(need[0])
    feat_names = np.asarray(img_db["img_name"])
    perm = np.argsort(feat_names)
    img_feats, feat_names = np.asarray(img_db["features"])[perm], feat_names[perm]
    start_idx = load(need[1])
    images, anns, cameras, joints, smpl = {}, [], {}, {}, {}
    for s in subjects:                                              # (the reference concatenates the subjects' lists in this order)
        d = load(osp.join(annot, f"Human36M_subject{s}_data.json"))
        images.update({im["id"]: im for im in d["images"]})
        anns += d["annotations"]
        cameras[str(s)] = load(osp.join(annot, f"Human36M_subject{s}_camera.json"))
        joints[str(s)] = load(osp.join(annot, f"Human36M_subject{s}_joint_3d.json"))
        smpl[str(s)] = load(osp.join(annot, f"Human36M_subject{s}_SMPL_NeuralAnnot.json"))
    rows, skipped = [], 0
    feat_cnt = -sampling_ratio
    for ann in {a["id"]: a for a in anns}.values():                 # pycocotools: a dict of annotation id, in file order
        img = images[ann["image_id"]]
        name = img["file_name"].split("/")[-1]
        frame = img["frame_idx"]
        if frame % sampling_ratio != 0:
            continue
        feat_cnt += sampling_ratio
        if name[:-12] == H36M_DROPPED_SEQUENCE:
            skipped += 1
            continue
        s, a, sa, c = str(img["subject"]), str(img["action_idx"]), str(img["subaction_idx"]), str(img["cam_idx"])
        cam = cameras[s][c]
        R, t, f, cc = (np.array(cam[k], dtype=np.float32) for k in ("R", "t", "f", "c"))
        fit = smpl[s].get(a, {}).get(sa, {}).get(str(frame))
        bbox = _sanitized_bbox(np.array(ann["bbox"], dtype=np.float32), aspect_ratio)
        if bbox is None:
            skipped += 1
            continue
        world = np.array(joints[s][a][sa][str(frame)], dtype=np.float32)
        jcam = np.dot(R, world.transpose(1, 0)).transpose(1, 0) + t.reshape(1, 3)                       # world2cam (coord_utils.py:136-138)
        jimg = np.concatenate(((jcam[:, 0] / jcam[:, 2] * f[0] + cc[0])[:, None], (jcam[:, 1] / jcam[:, 2] * f[1] + cc[1])[:, None],
                               jcam[:, 2][:, None]), 1)                                                 # cam2pixel (:128-133)
        if frame == 0:
            feat_cnt = start_idx[s][a][sa][c]
        if feat_names[feat_cnt].split("/")[-1] != name:
            raise ValueError(f"feature database out of step with the annotations: entry {feat_cnt} is {feat_names[feat_cnt]}, the frame is {name}")
        rows.append((name, (img["height"], img["width"]), np.asarray(img_feats[feat_cnt], dtype=np.float32), jcam.astype(np.float32),
                     jimg.astype(np.float32), bbox.astype(np.float32), img["cam_idx"], fit, f, cc, R, t))
    names = np.array([r[0] for r in rows])
    # the CPN detections: sorted by name, every sampling_ratio-th image index kept (dataset.py:110-130); the reference indexes them with the
    # dataset index and asserts the names - here they are matched by name, which is the same thing on consistent files and says so on others
    det = load(need[2])
    det = {k.split("/")[-1]: v for k, v in det.items() if (int(k[-10:-4]) - 1) % sampling_ratio == 0}
    absent = [n for n in names if n not in det]
    if absent:
        raise ValueError(f"{len(absent)} frames have no CPN detection (first: {absent[0]})")
    if list(np.sort(np.array(list(det)))) != list(names):
        raise ValueError("the detection file and the annotations do not list the same frames in the same (sorted) order - the reference's "
                         "index-aligned detection table would be out of step (its own assert, dataset.py:559-560)")
    keypoints = np.stack([np.asarray(det[n], dtype=np.float32) for n in names])
    col = lambda i, dt=np.float32: np.asarray([r[i] for r in rows], dtype=dt)
    valid = np.array([r[7] is not None for r in rows])
    z72, z10, z3 = [0.0] * 72, [0.0] * 10, [0.0] * 3
    return FrameTable(
        name=f"Human3.6M {split} (protocol {protocol})", img_paths=names, vid_names=np.array([n[:-11] for n in names]), img_shapes=col(1, np.int32),
        keypoints=keypoints, features=col(2), joints_cam_h36m=col(3), joints_cam_coco=np.zeros((len(rows), 0, 3), np.float32),
        gt_joints_img_coco=col(4),
        smpl={"pose": np.asarray([r[7]["pose"] if r[7] else z72 for r in rows], np.float32), "shape": np.asarray([r[7]["shape"] if r[7] else z10 for r in rows], np.float32),
              "trans": np.asarray([r[7]["trans"] if r[7] else z3 for r in rows], np.float32), "gender": np.array(["neutral"] * len(rows))},
        skipped=skipped, joints_name=H36M_JOINTS, extra_joints=0, mid_valid=valid, cam_idxs=col(6, np.int64),
        extras={"bboxs": col(5), "cam_focals": col(8), "cam_princpts": col(9), "cam_Rs": col(10), "cam_ts": col(11)})


# The 17 MPI-INF-3DHP test joints (lib/_kp_utils.py:46-65) as rows of the database's 49-joint SPIN order (:212-263), ``convert_kps(joints3D,
# "spin", "mpii3d_test")`` (data/MPII3D/dataset.py:270): headtop, neck, r/l shoulder-elbow-wrist, r/l hip-knee-ankle, hip, Spine (H36M), Head (H36M)
MPII3D_FROM_SPIN = (38, 37, 33, 32, 31, 34, 35, 36, 27, 26, 25, 28, 29, 30, 39, 41, 43)
# ... which the dataset names ('Head', 'Neck', 'R_Shoulder', ..., 'L_Ankle', 'Pelvis', 'Torso', 'Nose') (dataset.py:36-39), re-ordered into the
# Human3.6M joint set by ``transform_joint_to_other_db`` (:271; lib/aug_utils.py:10-21): H36M_JOINTS[k] = mpii3d joint MPII3D_TO_H36M[k]
MPII3D_TO_H36M = (14, 8, 9, 10, 11, 12, 13, 15, 1, 16, 0, 5, 6, 7, 2, 3, 4)


def load_mpii3d(data_path: str, split: str = "val") -> FrameTable:
    """What ``MPII3D.load_data_val`` reads (data/MPII3D/dataset.py:249-292; the validation split is the reference's test split, :24-25): the
    joblib database ``mpii3d_<split>_scale12_db.pt`` (image names, 2048-d features, 49 SPIN joints in metres) and the ViTPose output
    ``vitpose_mpii3d_<split>_output.json`` - sorted by image name; joints as the reference makes them (17 test joints -> Human3.6M order,
    fp32, x 1000 mm); every image 2048 x 2048.  The model input is the COCO set of 19 (config/test_mesh_mpii3d.yml: pelvis and neck appended
    by ``FrameTable.pose2d``), the windows are ``FrameTable.windows`` (``split_into_chunks_pose``, :103: no frame is filtered), the target of
    ``MPII3D.evaluate`` is ``FrameTable.gt_joints_root_relative`` (:482-484, 501)."""
    db_file = osp.join(data_path, f"mpii3d_{split}_scale12_db.pt")
    vit_file = osp.join(data_path, f"vitpose_mpii3d_{split}_output.json")
    missing = [p for p in (db_file, vit_file) if not osp.exists(p)]
    if missing:
        raise FileNotFoundError("MPI-INF-3DHP files missing under %s: %s" % (data_path, ", ".join(osp.basename(p) for p in missing)))
    import joblib
    db = joblib.load(db_file)
    det = {str(item["image_name"]): np.asarray(item["keypoints"], dtype=np.float32)[:, :3] for item in json.load(open(vit_file))}
    names = [str(n) for n in db["img_name"]]
    absent = [n for n in names if n not in det]
    if absent:
        raise ValueError(f"vitpose_mpii3d_{split}_output.json holds no detection for {len(absent)} image(s), first: {absent[0]}")
    j49 = np.asarray(db["joints3D"])
    # convert_kps copies the rows into a float64 table, transform_joint_to_other_db into a float32 one; the factor 1000 acts on the fp32 values
    joints = np.asarray(j49[:, MPII3D_FROM_SPIN, :3], dtype=np.float64)[:, MPII3D_TO_H36M].astype(np.float32) * 1000
    img_paths = np.array(names)
    perm = np.argsort(img_paths)
    img_paths = img_paths[perm]
    n = len(img_paths)
    return FrameTable(
        name=f"MPI-INF-3DHP {split}", img_paths=img_paths, vid_names=np.array([p[:-11] for p in img_paths]),
        img_shapes=np.full((n, 2), 2048, dtype=np.int32), keypoints=np.stack([det[p] for p in img_paths]).astype(np.float32),
        features=np.asarray(db["features"], dtype=np.float32)[perm], joints_cam_h36m=np.ascontiguousarray(joints[perm], dtype=np.float32),
        joints_cam_coco=np.zeros((n, 19, 3), np.float32), gt_joints_img_coco=np.zeros((n, 19, 3), np.float32))


def window_frames(win: np.ndarray, seqlen: int = 16) -> np.ndarray:
    """[W, seqlen] frame indices of windows given as inclusive (start, end) pairs; start == end repeats one frame (dataset.py:213-216)."""
    win = np.asarray(win).reshape(-1, 2)
    step = (win[:, 1] != win[:, 0]).astype(np.int64)
    return win[:, :1] + step[:, None] * np.arange(seqlen)[None, :]


def window_mid(win: np.ndarray, seqlen: int = 16) -> np.ndarray:
    """Frame index whose targets a window carries: its middle frame, or the one repeated frame of a start == end window
    (Human36M/dataset.py:737-740, PW3D/dataset.py:245-251) - the same rule as :func:`window_frames`."""
    win = np.asarray(win).reshape(-1, 2)
    return win[:, 0] + (seqlen // 2) * (win[:, 1] != win[:, 0])


def window_batch(pose2d_frames, feat_frames, win, seqlen: int = 16):
    """Model inputs of a batch of windows, gathered ON THE DEVICE from the per-frame tables (each 8 KB frame feature was uploaded
    once, not 16 times): (pose2d [W, seqlen, J, 2], img_feat [W, seqlen, 2048])."""
    import torch
    idx = torch.from_numpy(window_frames(win, seqlen)).to(pose2d_frames.device)
    return pose2d_frames[idx], feat_frames[idx]
