"""Experiment: does running two half-batches as two concurrent pipelines (two model handles, two streams) beat one full batch?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PMCE_SYNTHETIC_BASE_DATA", "1")   # synthetic weights on the synthetic template (explicit opt-in)
import torch
from pmce_amd import assets, models, synth
dev = torch.device("cuda:0")
J = 17
sd = synth.make_state_dict(synth.pmce_spec(J, 256, 3), seed=123)
def mk():
    m = models.PMCE.get_model(J, 256, 3); m.load_state_dict(sd); m.set_j_regressor(assets.load_j_regressor("h36m")); return m.to(dev)
def run(nstreams, B, steps=20):
    ms = [mk() for _ in range(nstreams)]
    ss = [torch.cuda.Stream() for _ in range(nstreams)]
    ins = []
    for i in range(nstreams):
        p, f = synth.make_inputs(B, J, seed=i)
        ins.append((torch.from_numpy(p).to(dev), torch.from_numpy(f).to(dev)))
    for r in range(3):
        for i in range(nstreams):
            with torch.cuda.stream(ss[i]): ms[i].forward_with_joints(*ins[i])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for r in range(steps):
        for i in range(nstreams):
            with torch.cuda.stream(ss[i]): ms[i].forward_with_joints(*ins[i])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{nstreams} pipeline(s) x B={B}: {nstreams * B * steps / dt:.0f} clips/s", flush=True)
    del ms
    torch.cuda.empty_cache()
run(1, 256); run(2, 128); run(2, 256); run(1, 512); run(4, 64); run(3, 256)
