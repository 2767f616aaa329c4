"""Small-batch A/B helper: host-observed latency of forward_with_joints + synchronize at B = 1 / 8 / 32 / 64 (p50 of 200 calls), and the decoder-only
forward at batch 64 (back-to-back, 200 calls) - for two builds of the library inside ONE gpurun call (PMCE_LIB_PATH).  Prints one JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("PMCE_SYNTHETIC_BASE_DATA", "1")
import torch
from pmce_amd import _lib, assets, models, synth

dev = torch.device("cuda:0")
J = 17
sd = synth.make_state_dict(synth.pmce_spec(J, 256, 3), seed=123)
model = models.PMCE.get_model(J, 256, 3)
model.load_state_dict(sd)
model.set_j_regressor(assets.load_j_regressor("h36m"))
model = model.to(dev)
model.set_overflow_policy("report")
out = {"lib": os.path.basename(_lib.LIB_PATH)}
for B in (1, 8, 32, 64):
    p = torch.rand(B, 16, J, 2, device=dev) * 2 - 1
    f = torch.relu(torch.randn(B, 16, 2048, device=dev))
    for _ in range(20):
        model.forward_with_joints(p, f)
    torch.cuda.synchronize()
    ts = []
    for _ in range(200):
        t0 = time.perf_counter()
        model.forward_with_joints(p, f)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    model.profile(True)
    for _ in range(5):
        model.forward_with_joints(p, f)
    torch.cuda.synchronize()
    prof = model.profile_read()
    model.profile(False)
    out[f"B{B}"] = {"p50_ms": round(ts[100], 4), "min_ms": round(ts[0], 4), "gru_step_us_per_forward": round(prof["gru_step"][0] / 5 * 1e3, 1)}
dec = models.CoevoDecoder.get_model(J, 256)
dec.load_state_dict({k[len("pose_mesh_coevo."):]: v for k, v in sd.items() if k.startswith("pose_mesh_coevo.")})
dec = dec.to(dev)
dec.set_overflow_policy("report")
joints = (torch.randn(64, J, 3) * 0.3).to(dev)
feats = torch.relu(torch.randn(64, 16, 2048)).to(dev)
for _ in range(20):
    dec(joints, feats)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(400):
    dec(joints, feats)
torch.cuda.synchronize()
out["decoder_b64_clips_s"] = round(64 * 400 / (time.perf_counter() - t0), 1)
print(json.dumps(out))
