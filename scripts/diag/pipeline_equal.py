"""Diagnostic: direct forwards vs pipeline lanes on the same batches; prints which batches / clips / outputs differ.
   python scripts/diag/pipeline_equal.py [C] [B]     (PMCE_SPLIT_OVERLAP=0 for the serial schedule)"""
import sys
import os.path as osp
sys.path.insert(0, osp.dirname(osp.dirname(osp.dirname(osp.abspath(__file__)))))
import torch
from pmce_amd import assets, models, synth

C_ = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 6
mode = sys.argv[3] if len(sys.argv) > 3 else "split_f16"
DEV = torch.device("cuda:0")
J = 17
model = models.PMCE.get_model(J, C_, 3)
model.load_state_dict(synth.make_state_dict(synth.pmce_spec(J, C_, 3), seed=123))
model.set_j_regressor(assets.load_j_regressor("h36m"))
model = model.to(DEV)
if mode != "split_f16":
    model.set_gemm_mode(mode)
batches = []
for i in range(7):
    p, f = synth.make_inputs(B, J, 900 + i)
    batches.append((torch.from_numpy(p).to(DEV), torch.from_numpy(f).to(DEV)))
want = [[o.clone() for o in model.forward_with_joints(p, f)] for p, f in batches]
again = [[o.clone() for o in model.forward_with_joints(p, f)] for p, f in batches]
names = ["mesh", "pose", "pose3d", "pred"]
print("direct forward twice: equal =", all(torch.equal(a, b) for w, g in zip(want, again) for a, b in zip(w, g)))
for depth in (1, 2, 3):
    for stagger in (True, False):
        pipe = model.pipeline(depth=depth, stagger=stagger)
        for rep in range(2):
            tickets = [pipe.submit(p, f) for p, f in batches]
            outs = [t.result() for t in tickets]
            torch.cuda.synchronize()
            bad = []
            for k, (w, g) in enumerate(zip(want, outs)):
                for n, a, b in zip(names, w, g):
                    if not torch.equal(a, b):
                        clips = (a != b).flatten(1).any(1).nonzero().flatten().tolist()
                        bad.append((k, n, clips, float((a - b).abs().max())))
            print(f"depth {depth} stagger {stagger} rep {rep}: {'all equal' if not bad else bad}")
        print("   overflowed:", [e.overflowed() for e in pipe.engines])
