"""Writer of a SMALL synthetic 3DPW test set in the reference's own file formats (data/PW3D/dataset.py:90-128 reads exactly these
six files) - data, not reference code.  Used by make_golden_datasets.py (the reference's PW3D class loads the directory) and by
tests/test_datasets_host.py (pmce_amd.datasets loads the same directory; the files are regenerated, not committed: 1.4 MB of JSON).

Layout on purpose: three sequences, one with two persons; annotations listed in SHUFFLED order (the loader must sort by image path);
one annotation without a feature entry (the reference skips it, dataset.py:148-151); one (sequence, person) shorter than 16 frames
(no window); image sizes that differ per sequence (portrait and landscape); keypoints with a confidence column."""
import json
import os
import os.path as osp

import numpy as np

SEQS = (("downtown_walk_00", (0, 1), 37, (1920, 1080)),      # name, person ids, frames, (height, width)
        ("office_phoneCall_00", (0,), 9, (1080, 1920)),
        ("outdoors_fencing_01", (0,), 52, (1080, 1920)))
MISSING_FEATURE = ("outdoors_fencing_01", 0, 30)             # (sequence, person, frame): no entry in the feature file


def write(root, split="test", seed=11):
    """Create <root>/PW3D/pw3d_data/*.json.  Returns the data path."""
    rng = np.random.default_rng(seed)
    path = osp.join(root, "PW3D", "pw3d_data")
    os.makedirs(path, exist_ok=True)
    images, anns = [], []
    coco_cam, gt_img, h36m_cam, feats, vit = {}, {}, {}, {}, []
    img_id = 0
    for seq, persons, n, (h, w) in SEQS:
        for fr in range(n):
            img_id += 1
            images.append({"id": img_id, "width": w, "height": h, "sequence": seq, "file_name": f"image_{fr:05d}.jpg"})
            for pid in persons:
                aid = img_id * 10 + pid
                anns.append({"id": aid, "image_id": img_id, "person_id": pid,
                             "smpl_param": {"pose": rng.normal(0, 0.3, 72).round(5).tolist(), "shape": rng.normal(0, 1, 10).round(5).tolist(),
                                            "trans": rng.normal(0, 1, 3).round(5).tolist(), "gender": "male" if pid == 0 else "female"}})
                s, i, p = seq, str(fr), str(pid)
                root_j = rng.normal(0, 500, (1, 3)) + np.array([[0, 0, 4000.0]])
                coco_cam.setdefault(s, {}).setdefault(i, {})[p] = (root_j + rng.normal(0, 250, (19, 3))).round(3).tolist()
                h36m_cam.setdefault(s, {}).setdefault(i, {})[p] = (root_j + rng.normal(0, 250, (17, 3))).round(3).tolist()
                kp = np.concatenate([rng.uniform(0, w, (17, 1)), rng.uniform(0, h, (17, 1)), rng.uniform(0.2, 1.0, (17, 1))], 1)
                gt_img.setdefault(s, {}).setdefault(i, {})[p] = np.concatenate([kp[:, :2] + rng.normal(0, 3, (17, 2)), np.ones((17, 1))], 1).round(3).tolist()
                vit.append({"annotation_id": aid, "image_id": img_id, "keypoints": np.concatenate([kp, np.zeros((17, 1))], 1).round(3).tolist()})
                if (seq, pid, fr) != MISSING_FEATURE:
                    feats[f"{seq}_{pid}_{fr}"] = np.maximum(rng.normal(0, 1, 2048), 0).round(4).tolist()
    order = rng.permutation(len(anns))
    anns = [anns[k] for k in order]
    rng.shuffle(vit)
    dump = lambda name, obj: json.dump(obj, open(osp.join(path, name), "w"))
    dump(f"3DPW_latest_{split}.json", {"images": images, "annotations": anns})
    dump(f"vitpose_3dpw_{split}_output.json", vit)
    dump(f"3DPW_{split}_joint_coco_cam.json", coco_cam)
    dump(f"3DPW_{split}_gt_joint_coco_img.json", gt_img)
    dump(f"3DPW_{split}_joint_h36m_cam.json", h36m_cam)
    dump(f"3DPW_{split}_img_feat.json", feats)
    return path
