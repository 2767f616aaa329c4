import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("PMCE_SYNTHETIC_BASE_DATA", "1")   # synthetic weights on the synthetic template (explicit opt-in)
import numpy as np, torch
from pmce_amd import assets, models, synth, staging
dev = torch.device("cuda:0"); B, J, C = 256, 17, 256
sd = synth.make_state_dict(synth.pmce_spec(J, C, 3), seed=123)
model = models.PMCE.get_model(J, C, 3); model.load_state_dict(sd); model.set_j_regressor(assets.load_j_regressor("h36m")); model = model.to(dev)
p_np, f_np = synth.make_inputs(B, J, seed=1)
p, f = torch.from_numpy(p_np).to(dev), torch.from_numpy(f_np).to(dev)
for _ in range(3): model.forward_with_joints(p, f)
torch.cuda.synchronize()
def timeit(name, fn, n=10):
    torch.cuda.synchronize(); t = time.perf_counter(); fn(n); torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter()-t)/n*1e3:.2f} ms/step", flush=True)
timeit("resident", lambda n: [model.forward_with_joints(p, f) for _ in range(n)])
bufs = [(p.clone(), f.clone()) for _ in range(3)]
timeit("rotating device buffers", lambda n: [model.forward_with_joints(*bufs[i % 3]) for i in range(n)])
feeder = staging.PinnedFeeder(dev, {"pose2d": ((B, 16, J, 2), torch.float32), "img_feat": ((B, 16, 2048), torch.float32)}, slots=3)
hb = lambda n: [{"pose2d": p_np, "img_feat": f_np}] * n
import time as _t
t0=_t.perf_counter(); feeder.host[0]["img_feat"].copy_(torch.from_numpy(f_np)); print("host copy img_feat ms", (_t.perf_counter()-t0)*1e3)
t0=_t.perf_counter(); feeder.host[0]["img_feat"][:B].copy_(torch.as_tensor(f_np)); print("host copy slice ms", (_t.perf_counter()-t0)*1e3)
t0=_t.perf_counter(); feeder.host[1]["img_feat"][:B].copy_(torch.as_tensor(f_np)); print("host copy slot1 ms", (_t.perf_counter()-t0)*1e3)
t0=_t.perf_counter(); feeder.host[1]["img_feat"][:B].copy_(torch.as_tensor(f_np)); print("host copy slot1 again ms", (_t.perf_counter()-t0)*1e3)
timeit("feeder only", lambda n: [d for d in feeder.run(hb(n))])
timeit("feeder + forward", lambda n: [model.forward_with_joints(d["pose2d"], d["img_feat"]) for d in feeder.run(hb(n))])
def manual(n):
    cs = torch.cuda.Stream()
    pin = [(torch.empty_like(p, device="cpu").pin_memory(), torch.empty_like(f, device="cpu").pin_memory()) for _ in range(2)]
    ev = [torch.cuda.Event() for _ in range(2)]
    for i in range(n):
        s = i % 2
        np.copyto(pin[s][0].numpy(), p_np); np.copyto(pin[s][1].numpy(), f_np)
        with torch.cuda.stream(cs):
            bufs[s][0].copy_(pin[s][0], non_blocking=True); bufs[s][1].copy_(pin[s][1], non_blocking=True); ev[s].record(cs)
        torch.cuda.current_stream().wait_event(ev[s])
        model.forward_with_joints(*bufs[s])
timeit("manual copy-stream (no slot sync)", manual)
