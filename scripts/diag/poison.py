"""Diagnostic: does any kernel read workspace / arena memory that nobody wrote?  The workspace is filled with NaN (then with large
finite values) between two forwards of the same batch; a second model is built after gigabytes of poisoned memory were returned
to PyTorch's allocator.   python scripts/diag/poison.py [C] [B]"""
import sys
import os.path as osp
sys.path.insert(0, osp.dirname(osp.dirname(osp.dirname(osp.abspath(__file__)))))
import torch
from pmce_amd import assets, models, synth

C_ = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 6
DEV = torch.device("cuda:0")
J = 17
names = ["mesh", "pose", "pose3d", "pred"]


def build():
    m = models.PMCE.get_model(J, C_, 3)
    m.load_state_dict(synth.make_state_dict(synth.pmce_spec(J, C_, 3), seed=123))
    m.set_j_regressor(assets.load_j_regressor("h36m"))
    return m.to(DEV)


def report(tag, want, got):
    bad = []
    for n, a, b in zip(names, want, got):
        if not torch.equal(a, b):
            clips = ((a != b) | torch.isnan(b)).flatten(1).any(1).nonzero().flatten().tolist()
            bad.append((n, clips, float((a - b).abs().nan_to_num(1e30).max())))
    print(f"{tag}: {'equal' if not bad else bad}", flush=True)


p, f = synth.make_inputs(B, J, 900)
p, f = torch.from_numpy(p).to(DEV), torch.from_numpy(f).to(DEV)
for mode in ("split_f16", "f32"):
    model = build()
    if mode == "f32":
        model.set_gemm_mode("f32")
    want = [o.clone() for o in model.forward_with_joints(p, f)]
    eng = model._ensure_packed()
    for what, val in (("nan", float("nan")), ("1e30", 1e30), ("-3", -3.0)):
        eng.ws.view(torch.float32).fill_(val)
        report(f"[{mode}] workspace filled with {what}", want, model.forward_with_joints(p, f))
    for what, val in (("nan", float("nan")), ("1e4", 1e4)):
        junk = [torch.full((1 << 28,), val, device=DEV) for _ in range(4)]   # 4 GB
        del junk
        m2 = build()
        if mode == "f32":
            m2.set_gemm_mode("f32")
        report(f"[{mode}] new model after 4 GB of {what} went back to the allocator", want, m2.forward_with_joints(p, f))
        pipe = m2.pipeline(depth=2)
        outs = [pipe.submit(p, f).result() for _ in range(3)]
        for k, o in enumerate(outs):
            report(f"[{mode}]    its pipeline, submit {k}", want, o)
        del m2, pipe
