"""Does the FIRST model of a process run its kernels slower than a later one?  (bench.py's second record, when it was measured
in the same process after the headline, showed 9 % faster GEMMs in the profiling pass than any fresh process.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("PMCE_SYNTHETIC_BASE_DATA", "1")
import torch
from pmce_amd import assets, models, synth

dev = torch.device("cuda:0")
J, B = 17, 256
order = [int(c) for c in (sys.argv[1:] or ["256", "256", "512", "256"])]
sds = {}
for n, C in enumerate(order):
    if C not in sds:
        sds[C] = synth.make_state_dict(synth.pmce_spec(J, C, 3), seed=123)
    m = models.PMCE.get_model(J, C, 3)
    m.load_state_dict(sds[C])
    m.set_j_regressor(assets.load_j_regressor("h36m"))
    m = m.to(dev)
    p = torch.rand(B, 16, J, 2, device=dev) * 2 - 1
    f = torch.relu(torch.randn(B, 16, 2048, device=dev))
    for _ in range(5):
        m.forward_with_joints(p, f)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        m.forward_with_joints(p, f)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    m.profile(True)
    for _ in range(3):
        m.forward_with_joints(p, f)
    torch.cuda.synchronize()
    prof = m.profile_read()
    m.profile(False)
    k = {a: round(v[0] / 3, 3) for a, v in prof.items() if v[1] > 0}
    print(f"model #{n} C={C}: sequential forward {dt*1e3:.3f} ms; profile pass: gemm_lifter {k['gemm_lifter']} gemm_gru_in {k['gemm_gru_in']} "
          f"gru_step {k['gru_step']} vertex_sa {k['vertex_sa']} ln_chain {k['ln_chain']} sum {sum(k.values()):.3f}", flush=True)
    del m, p, f
    torch.cuda.empty_cache()
