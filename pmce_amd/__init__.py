"""pmce_amd — MI355X-native (gfx950) implementation of kasvii/PMCE's per-clip inference hot path.

    from pmce_amd import models
    model = models.PMCE.get_model(num_joint=17, embed_dim=256, depth=3).cuda()
    model.load_state_dict(torch.load(ckpt)['model_state_dict'])
    cam_mesh, cam_pose, pose3d = model(pose2d, img_feat)

The arithmetic lives in libpmce_hip.so (pmce_amd/csrc, C ABI in include/pmce_hip.h)."""
__version__ = "0.1.0"

from . import config  # noqa: F401


def j_regress(cam_mesh_m, j_regressor):
    """Drop-in for ``torch.matmul(J_regressor[None], pred_mesh*1000)`` (lib/core/base.py:223-225) on the GPU."""
    from . import ops
    return ops.j_regress(cam_mesh_m, j_regressor)
