"""Writer of a SMALL synthetic MPI-INF-3DHP validation set in the reference's own file formats (data/MPII3D/dataset.py:249-292 reads exactly
these two files: the joblib database ``mpii3d_val_scale12_db.pt`` and the ViTPose output ``vitpose_mpii3d_val_output.json``) - data, not
reference code.  Used by make_golden_datasets_mpii3d.py (the reference's MPII3D class loads the directory) and by tests/test_datasets_host.py
(pmce_amd.datasets loads the same directory; the files are regenerated, not committed).

Layout on purpose: three videos (image names end in ``_NNNNNN.jpg``: the video is the name without its last 11 characters,
lib/_img_utils.py:31), one shorter than 16 frames (no window); database rows in SHUFFLED order (the loader sorts by image name); joints in
the 49-joint SPIN order (lib/_kp_utils.py:212-263), metres; detector keypoints with a confidence column."""
import json
import os
import os.path as osp

import numpy as np

VIDEOS = (("TS1/imageSequence/img", 53), ("TS3/imageSequence/img", 11), ("TS6/imageSequence/img", 37))


def write(root, split="val", seed=29):
    """Create <root>/MPII3D/mpii3d_data/{mpii3d_<split>_scale12_db.pt, vitpose_mpii3d_<split>_output.json}.  Returns the data path."""
    import joblib
    rng = np.random.default_rng(seed)
    path = osp.join(root, "MPII3D", "mpii3d_data")
    os.makedirs(path, exist_ok=True)
    names, feats, joints, vit = [], [], [], []
    for vid, n in VIDEOS:
        for fr in range(1, n + 1):
            name = f"{vid}_{fr:06d}.jpg"
            names.append(name)
            feats.append(np.maximum(rng.normal(0, 1, 2048), 0).astype(np.float32))
            root_j = rng.normal(0, 0.5, (1, 3)) + np.array([[0, 0, 4.0]])
            joints.append((root_j + rng.normal(0, 0.25, (49, 3))).astype(np.float32))          # metres, SPIN order
            kp = np.concatenate([rng.uniform(0, 2048, (17, 2)), rng.uniform(0.2, 1.0, (17, 1)), np.zeros((17, 1))], 1)
            vit.append({"image_name": name, "keypoints": kp.round(3).tolist()})
    order = rng.permutation(len(names))
    db = {"img_name": np.array(names)[order], "features": np.stack(feats)[order], "joints3D": np.stack(joints)[order]}
    joblib.dump(db, osp.join(path, f"mpii3d_{split}_scale12_db.pt"))
    rng.shuffle(vit)
    json.dump(vit, open(osp.join(path, f"vitpose_mpii3d_{split}_output.json"), "w"))
    return path
