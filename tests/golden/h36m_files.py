"""Writer of a SMALL synthetic Human3.6M test set in the reference's own file formats (data/Human36M/dataset.py:194-269 and :105-130 read
exactly these files) - data, not reference code.  Used by make_golden_datasets_h36m.py (the reference's Human36M class loads the
directory) and by tests/test_datasets_host.py (pmce_amd.datasets.load_h36m loads the same directory; the files are regenerated, not
committed: 3 MB).

Layout on purpose (protocol 2 test subjects 9 and 11; every second frame is kept): a video shorter than 16 kept frames (no window); the
sequence the reference drops by name ('s_11_act_02_subact_02_ca_0*'); frames without an SMPL fit (windows whose middle frame is one of them
are dropped, lib/_img_utils.py:75); one annotation with an empty bounding box (dropped, dataset.py:286-287); cameras other than 4 (the
evaluation keeps camera 4 only, dataset.py:742-744); two image sizes; a feature database that holds EVERY frame (the loader walks it with
the start index of each video, dataset.py:301-305) and a detection file that holds every frame but the dropped ones."""
import json
import os
import os.path as osp

import numpy as np

# (subject, action, subaction, camera, frames)
VIDEOS = ((9, 2, 1, 1, 70), (9, 2, 1, 4, 80), (9, 3, 2, 4, 21), (11, 2, 2, 1, 36), (11, 5, 1, 4, 140))
IMG_HW = {9: (1002, 1000), 11: (1000, 1000)}
NO_SMPL = {(9, 2, 1): (22, 24, 25), (11, 5, 1): tuple(range(60, 66))}     # (subject, action, subaction) -> frames without a fit
EMPTY_BBOX = (9, 2, 1, 1, 6)                                               # this frame's annotation has w = 0
DROPPED_PREFIX = "s_11_act_02_subact_02_ca_0"


def img_name(s, a, sa, c, frame):
    return f"s_{s:02d}_act_{a:02d}_subact_{sa:02d}_ca_{c:02d}_{frame + 1:06d}.jpg"


def write(root, split="test", seed=23):
    """Create <root>/Human36M/h36m_data/{annotations/*.json, *.json, *.pt}.  Returns the data path."""
    import joblib
    rng = np.random.default_rng(seed)
    path = osp.join(root, "Human36M", "h36m_data")
    annot = osp.join(path, "annotations")
    os.makedirs(annot, exist_ok=True)
    dump = lambda p, obj: json.dump(obj, open(p, "w"))
    all_names, det = [], {}
    img_id = 0
    for subject in (9, 11):
        images, anns, joints, smpl = [], [], {}, {}
        cams = {}
        for c in (1, 2, 3, 4):
            q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
            cams[str(c)] = {"R": (q * np.sign(np.linalg.det(q))).round(6).tolist(), "t": (rng.normal(0, 300, 3) + [0, 0, 4500]).round(3).tolist(),
                            "f": rng.uniform(1100, 1200, 2).round(3).tolist(), "c": rng.uniform(480, 520, 2).round(3).tolist()}
        for (s, a, sa, c, n) in VIDEOS:
            if s != subject:
                continue
            h, w = IMG_HW[s]
            for fr in range(n):
                img_id += 1
                name = img_name(s, a, sa, c, fr)
                all_names.append(name)
                images.append({"id": img_id, "file_name": name[:-11] + "/" + name, "width": w, "height": h, "subject": s, "action_idx": a,
                               "subaction_idx": sa, "cam_idx": c, "frame_idx": fr})
                bw = 0.0 if (s, a, sa, c, fr) == EMPTY_BBOX else float(rng.uniform(150, 400))
                anns.append({"id": img_id, "image_id": img_id, "bbox": [float(rng.uniform(100, 400)), float(rng.uniform(100, 300)), bw, float(rng.uniform(300, 600))]})
                world = rng.normal(0, 400, (1, 3)) + rng.normal(0, 250, (17, 3))
                joints.setdefault(str(a), {}).setdefault(str(sa), {})[str(fr)] = world.round(3).tolist()     # (the same frame of another camera overwrites: one world pose per frame)
                if fr not in NO_SMPL.get((s, a, sa), ()):
                    smpl.setdefault(str(a), {}).setdefault(str(sa), {})[str(fr)] = {"pose": rng.normal(0, 0.3, 72).round(5).tolist(),
                                                                                     "shape": rng.normal(0, 1, 10).round(5).tolist(),
                                                                                     "trans": rng.normal(0, 1, 3).round(5).tolist()}
                if not name.startswith(DROPPED_PREFIX) and (s, a, sa, c, fr) != EMPTY_BBOX:
                    det[name] = np.concatenate([rng.uniform(0, w, (17, 1)), rng.uniform(0, h, (17, 1))], 1).round(3).tolist()
        dump(osp.join(annot, f"Human36M_subject{subject}_data.json"), {"images": images, "annotations": anns})
        dump(osp.join(annot, f"Human36M_subject{subject}_camera.json"), cams)
        dump(osp.join(annot, f"Human36M_subject{subject}_joint_3d.json"), joints)
        dump(osp.join(annot, f"Human36M_subject{subject}_SMPL_NeuralAnnot.json"), smpl)
    # the feature database: every frame, in any order (the loader sorts by name); the start index of every video in the SORTED list
    order = rng.permutation(len(all_names))
    names = np.array(all_names)[order]
    feats = np.maximum(rng.normal(0, 1, (len(names), 2048)), 0).astype(np.float32)
    joblib.dump({"features": feats, "img_name": names}, osp.join(path, f"h36m_{split}_imgfeat_db_concat.pt"))
    sorted_names = np.sort(names)
    start = {}
    for (s, a, sa, c, n) in VIDEOS:
        start.setdefault(str(s), {}).setdefault(str(a), {}).setdefault(str(sa), {})[str(c)] = int(np.searchsorted(sorted_names, img_name(s, a, sa, c, 0)))
    dump(osp.join(path, f"Human36M_{split}_start_idx_tight.json"), start)
    keys = list(det)
    rng.shuffle(keys)
    dump(osp.join(path, f"Human36M_{split}_cpn_joint_2d.json"), {k: det[k] for k in keys})
    return path
