"""Pipeline depth x staggering x in-forward two-stream mode, interleaved rounds in one process (GPU only)."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("PMCE_SYNTHETIC_BASE_DATA", "1")   # synthetic weights on the synthetic template (explicit opt-in)
import torch
from pmce_amd import assets, models, synth
dev = torch.device("cuda:0"); B, J = 256, 17
sd = synth.make_state_dict(synth.pmce_spec(J, 256, 3), seed=123)
model = models.PMCE.get_model(J, 256, 3); model.load_state_dict(sd); model.set_j_regressor(assets.load_j_regressor("h36m")); model = model.to(dev)
p, f = (torch.from_numpy(a).to(dev) for a in synth.make_inputs(B, J, seed=1))
cfgs = [(1, False, True), (1, False, False), (2, False, True), (2, False, False), (2, True, True), (2, True, False), (3, False, True)]
if len(sys.argv) > 1:   # e.g. "2,1,1" = depth 2, stagger, two-stream forwards: only that configuration in this process
    d, st, cc = (int(x) for x in sys.argv[1].split(","))
    cfgs = [(d, bool(st), bool(cc))]
pipes = {c: model.pipeline(c[0], stagger=c[1]) for c in cfgs}
res = {c: [] for c in cfgs}
for rnd in range(5):
    for c in cfgs:
        pipe = pipes[c]
        for e in pipe.engines: e.set_concurrency(c[2])
        for _ in range(4): pipe.submit(p, f)
        torch.cuda.synchronize(); t = time.perf_counter()
        n = 24
        for _ in range(n): pipe.submit(p, f)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        res[c].append(B * n / dt)
for c in cfgs:
    r = res[c]
    print(f"depth {c[0]} stagger {c[1]!s:5} two-stream-forward {c[2]!s:5}: median {statistics.median(r):8.0f}  min {min(r):8.0f}  max {max(r):8.0f} clips/s", flush=True)
