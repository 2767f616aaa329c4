"""Latency of one forward at small batch sizes with the large products on the f16 matrix pipe (three-product form) vs the fp32 pipe,
both with the two-stream schedule: where should pmce_model_set_split_min_batch sit?   python scripts/microbench/small_batch_modes.py [C]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pmce_amd import assets, models, synth
C = int(sys.argv[1]) if len(sys.argv) > 1 else 512
J = 17
dev = torch.device("cuda:0")
assets.allow_synthetic_base_data()
model = models.PMCE.get_model(J, C, 3)
model.load_state_dict(synth.make_state_dict(synth.pmce_spec(J, C, 3), seed=123))
model.set_j_regressor(assets.load_j_regressor("h36m"))
model = model.to(dev)
for B in (1, 2, 4, 8, 16, 24, 32, 48, 64, 96, 128):
    p = torch.rand(B, 16, J, 2, device=dev) * 2 - 1
    f = torch.relu(torch.randn(B, 16, 2048, device=dev))
    row = []
    for mode, mb in (("split_f16", 1), ("f32", None)):
        model.set_gemm_mode(mode, mb)
        for _ in range(5): model.forward_with_joints(p, f)
        torch.cuda.synchronize()
        ts = []
        for _ in range(40):
            t0 = time.perf_counter(); model.forward_with_joints(p, f); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        ts.sort(); row.append(ts[len(ts) // 2] * 1e3)
    print(f"C={C} B={B:4d}: split_f16 {row[0]:.3f} ms   f32 {row[1]:.3f} ms", flush=True)
