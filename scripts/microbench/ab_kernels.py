"""Per-kernel-class times (HIP events around every launch, one stream) of a forward at the batch sizes given, for A/B runs of two
builds of the library inside ONE gpurun call:
    for L in a b a b; do PMCE_LIB_PATH=pmce_amd/variants/libpmce_hip_$L.so python scripts/microbench/ab_kernels.py 1 256; done
"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("PMCE_SYNTHETIC_BASE_DATA", "1")
import torch
from pmce_amd import _lib, assets, models, synth

C = int(os.environ.get("AB_C", "256"))
J = 17
dev = torch.device("cuda:0")
model = models.PMCE.get_model(J, C, 3)
model.load_state_dict(synth.make_state_dict(synth.pmce_spec(J, C, 3), seed=123))
model.set_j_regressor(assets.load_j_regressor("h36m"))
model = model.to(dev)
model.set_overflow_policy("report")          # asynchronous calls (the default "rerun" policy waits for every forward)
only = os.environ.get("AB_CLASSES", "").split(",") if os.environ.get("AB_CLASSES") else None
out = {"lib": os.path.basename(_lib.LIB_PATH), "C": C}
for B in [int(a) for a in sys.argv[1:]] or [1, 256]:
    p = torch.rand(B, 16, J, 2, device=dev) * 2 - 1
    f = torch.relu(torch.randn(B, 16, 2048, device=dev))
    for _ in range(5):
        model.forward_with_joints(p, f)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(20):
        model.forward_with_joints(p, f)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 20
    model.profile(True)
    n = 10
    for _ in range(n):
        model.forward_with_joints(p, f)
    torch.cuda.synchronize()
    prof = model.profile_read()
    model.profile(False)
    out[f"B{B}"] = {k: round(v[0] / n * 1e3, 1) for k, v in prof.items() if v[1] > 0 and (only is None or k in only)}
    out[f"B{B}"]["wall_us"] = round(wall * 1e6, 1)          # un-instrumented forwards back to back (side streams, graphs as configured)
    out[f"B{B}"]["sum_us"] = round(sum(v[0] for v in prof.values()) / n * 1e3, 1)
print(json.dumps(out))
