"""Check that a forward (including its internal fork/join onto the side stream) can be captured in a HIP graph and replayed."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PMCE_SYNTHETIC_BASE_DATA", "1")   # synthetic weights on the synthetic template (explicit opt-in)
import torch
from pmce_amd import assets, models, synth
dev = torch.device("cuda:0")
J, B = 17, 256
model = models.PMCE.get_model(J, 256, 3)
model.load_state_dict(synth.make_state_dict(synth.pmce_spec(J, 256, 3), seed=123))
model.set_j_regressor(assets.load_j_regressor("h36m"))
model = model.to(dev)
p, f = synth.make_inputs(B, J, 0)
p, f = torch.from_numpy(p).to(dev), torch.from_numpy(f).to(dev)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): ref = model.forward_with_joints(p, f)      # warm-up: packing, workspace, side stream, attributes
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    out = model.forward_with_joints(p, f)
g.replay(); torch.cuda.synchronize()
print("graph replay max |diff| vs eager:", max(float((a - b).abs().max()) for a, b in zip(out, ref)))
for name, fn in (("eager", lambda: model.forward_with_joints(p, f)), ("graph replay", g.replay)):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
    print(f"{name}: {dt*1e3:.3f} ms/step  {B/dt:.0f} clips/s")
