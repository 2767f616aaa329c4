"""Is it safe to overlap kernels of the split-f16 mode once NO kernel of the library contains packed-fp32 arithmetic?
Every kernel of the path is deterministic, so outputs of overlapped runs (two streams inside a forward, PMCE_SPLIT_OVERLAP=1, and two
pipeline lanes on separate streams) must be BITWISE equal to the serial run's on the same inputs.  Prints mismatch counts and rates.
    PMCE_SPLIT_OVERLAP=1 python scripts/microbench/overlap_stress.py [C] [forwards]"""
import os, sys, time
os.environ.setdefault("PMCE_SPLIT_OVERLAP", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pmce_amd import assets, models, synth

C = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
J, B = 17, 256
dev = torch.device("cuda:0")
assets.allow_synthetic_base_data()
model = models.PMCE.get_model(J, C, 3)
model.load_state_dict(synth.make_state_dict(synth.pmce_spec(J, C, 3), seed=123))
model.set_j_regressor(assets.load_j_regressor("h36m"))
model = model.to(dev)
model.set_gemm_mode("split_f16")
gen = torch.Generator(device=dev); gen.manual_seed(7)
inputs = [(torch.rand(B, 16, J, 2, device=dev, generator=gen) * 2 - 1, torch.relu(torch.randn(B, 16, 2048, device=dev, generator=gen)))
          for _ in range(3)]

def run(p, f):
    return [None if t is None else t.clone() for t in model.forward_with_joints(p, f)]

model.set_concurrency(False)
refs = [run(p, f) for p, f in inputs]
again = [run(p, f) for p, f in inputs]
print("serial vs serial bitwise:", all(torch.equal(a, b) for r, s in zip(refs, again) for a, b in zip(r, s) if a is not None))
torch.cuda.synchronize(); t0 = time.time()
for i in range(20): model.forward_with_joints(*inputs[i % 3])
torch.cuda.synchronize(); print(f"serial: {(time.time() - t0) / 20 * 1e3:.3f} ms per forward")

model.set_concurrency(True)
bad = 0; worst = 0.0
for i in range(N):
    out = model.forward_with_joints(*inputs[i % 3])
    for a, b in ((x, y) for x, y in zip(out, refs[i % 3]) if x is not None):
        if not torch.equal(a, b):
            bad += 1; worst = max(worst, float((a - b).abs().max()))
torch.cuda.synchronize(); t0 = time.time()
for i in range(20): model.forward_with_joints(*inputs[i % 3])
torch.cuda.synchronize(); print(f"two streams inside a forward: {(time.time() - t0) / 20 * 1e3:.3f} ms per forward; "
                                f"{bad} of {N * len(refs[0])} outputs differ from the serial run (max abs {worst:.2e})")

pipe = model.pipeline(2).prepare(B)
print("lanes on", len({id(s) for s in pipe.streams}), "streams")
bad = 0; worst = 0.0
tickets = []
for i in range(N):
    tickets.append((i % 3, pipe.submit(*inputs[i % 3])))
    if len(tickets) >= 4:
        k, t = tickets.pop(0)
        for a, b in ((x, y) for x, y in zip(t.result(), refs[k]) if x is not None):
            if not torch.equal(a, b):
                bad += 1; worst = max(worst, float((a - b).abs().max()))
for k, t in tickets:
    for a, b in ((x, y) for x, y in zip(t.result(), refs[k]) if x is not None):
        if not torch.equal(a, b):
            bad += 1; worst = max(worst, float((a - b).abs().max()))
torch.cuda.synchronize(); t0 = time.time()
for i in range(40): pipe.submit(*inputs[i % 3])
torch.cuda.synchronize(); print(f"two lanes + two streams each: {(time.time() - t0) / 40 * 1e3:.3f} ms per forward; "
                                f"{bad} of {N * len(refs[0])} outputs differ from the serial run (max abs {worst:.2e})")
