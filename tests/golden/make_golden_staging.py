#!/usr/bin/env python3
"""Golden vectors for the input-staging functions from the REAL reference code (build container only):
PW3D.add_pelvis_and_neck / PW3D.normalize_screen_coordinates (data/PW3D/dataset.py:185-204, called as unbound methods) and
split_into_chunks_mesh (lib/_img_utils.py:58-92; skimage is absent, so `skimage.util.shape.view_as_windows` is provided by
numpy's sliding_window_view; cv2 / torchvision are imported by that module but not used by this function).  Inputs are regenerated from pmce_amd.synth by the tests; only outputs are stored."""
import os.path as osp, sys, types
import numpy as np

HERE = osp.dirname(osp.abspath(__file__)); REPO = osp.dirname(osp.dirname(HERE)); REF = "/root/reference"
sys.path.insert(0, REPO)
from pmce_amd import synth  # noqa: E402


def shims():
    class AD(dict):
        __getattr__ = dict.__getitem__
    core = types.ModuleType("core"); cc = types.ModuleType("core.config")
    cc.cfg = AD(data_dir="data", TEST=AD(vis=False), vis_dir="/tmp")
    core.config = cc
    pc = types.ModuleType("pycocotools"); pcc = types.ModuleType("pycocotools.coco"); pcc.COCO = object; pc.coco = pcc
    fu = types.ModuleType("funcs_utils"); fu.save_obj = lambda *a, **k: None
    sm = types.ModuleType("smpl"); sm.SMPL = object
    sk = types.ModuleType("skimage"); sku = types.ModuleType("skimage.util"); skus = types.ModuleType("skimage.util.shape")
    def view_as_windows(arr, window_shape, step=1):
        return np.lib.stride_tricks.sliding_window_view(arr, window_shape)[::step]
    skus.view_as_windows = view_as_windows; sku.shape = skus; sk.util = sku
    cv2 = types.ModuleType("cv2")
    tv = types.ModuleType("torchvision"); tvt = types.ModuleType("torchvision.transforms"); tv.transforms = tvt   # imported, unused here
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt})
    sys.modules.update({"core": core, "core.config": cc, "pycocotools": pc, "pycocotools.coco": pcc, "funcs_utils": fu,
                        "smpl": sm, "skimage": sk, "skimage.util": sku, "skimage.util.shape": skus, "cv2": cv2})
    sys.path.insert(0, osp.join(REF, "lib")); sys.path.insert(0, osp.join(REF, "data"))


def inputs():
    L = 40
    kp = synth.uniform_pm1("staging.kp", L * 17 * 3, 7).reshape(L, 17, 3).astype(np.float32)
    kp[..., 0] = (kp[..., 0] + 1) * 960.0     # x pixels
    kp[..., 1] = (kp[..., 1] + 1) * 540.0     # y pixels
    shapes = np.array([[1080, 1920] if i % 3 else [1920, 1080] for i in range(L)], dtype=np.int32)   # (height, width)
    return kp, shapes


def video_layout():
    """three videos (37, 9 and 64 frames) of two persons; frame 20 of the first has an invalid middle (pose param of length 1)."""
    names, valid = [], []
    for person, seq, n in ((0, "downtown_walk_00", 37), (0, "office_01", 9), (1, "downtown_walk_00", 64)):
        for i in range(n):
            names.append(f"{person}/{seq}/image_{i:05d}.jpg")
            valid.append(not (seq == "downtown_walk_00" and person == 0 and i == 20))
    return names, np.array(valid)


SINGLE_VIDEO_LENGTHS = (15, 16, 17, 31, 32, 33, 47, 48, 100, 1000)


def main():
    shims()
    from PW3D.dataset import PW3D
    import _img_utils
    kp, shapes = inputs()
    names = ('Nose', 'L_Eye', 'R_Eye', 'L_Ear', 'R_Ear', 'L_Shoulder', 'R_Shoulder', 'L_Elbow', 'R_Elbow', 'L_Wrist',
             'R_Wrist', 'L_Hip', 'R_Hip', 'L_Knee', 'R_Knee', 'L_Ankle', 'R_Ankle')
    ext = np.stack([PW3D.add_pelvis_and_neck(None, kp[i], names) for i in range(len(kp))])                 # [L,19,3]
    ext_p = np.stack([PW3D.add_pelvis_and_neck(None, kp[i], names, only_pelvis=True) for i in range(len(kp))])
    norm = np.stack([np.array(PW3D.normalize_screen_coordinates(None, ext[i][:, :2], w=shapes[i][1], h=shapes[i][0]), dtype=np.float32)
                     for i in range(len(kp))])                                                                  # as __getitem__ does
    img_names, valid = video_layout()
    poses = [np.zeros(72) if v else np.zeros(1) for v in valid]
    out = {"ext": ext, "ext_pelvis": ext_p, "norm": norm}
    for tag, (seqlen, stride, mv) in {"s1": (16, 1, True), "s16": (16, 16, True), "s4": (16, 4, True), "s1_nov": (16, 1, False)}.items():
        out["win_" + tag] = np.asarray(_img_utils.split_into_chunks_mesh(img_names, seqlen, stride, poses, is_train=False, match_vibe=mv)).reshape(-1, 2)
    # split_into_chunks_pose (lib/_img_utils.py:27-55: the lifter-only / streaming window list, no validity filter): the same
    # three-video layout, and single videos of several lengths (what pmce_amd.streaming.window_indices serves)
    for tag, (seqlen, stride, mv) in {"s1": (16, 1, True), "s16": (16, 16, True), "s4": (16, 4, True), "s1_nov": (16, 1, False)}.items():
        out["pose_win_" + tag] = np.asarray(_img_utils.split_into_chunks_pose(img_names, seqlen, stride, is_train=False, match_vibe=mv)).reshape(-1, 2)
    for L in SINGLE_VIDEO_LENGTHS:
        one = [f"0/only/image_{i:05d}.jpg" for i in range(L)]
        out[f"pose_win_single_{L}"] = np.asarray(_img_utils.split_into_chunks_pose(one, 16, 1, is_train=False, match_vibe=True)).reshape(-1, 2)
    np.savez_compressed(osp.join(HERE, "staging.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
