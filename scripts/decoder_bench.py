#!/usr/bin/env python3
"""BASELINE configs[1]: CoEvoDecoder-only forward (models.CoevoDecoder.get_model), batch 64, one MI355X.  Prints clips/s, the
north-star cross-attention kernel against the HBM roof (the CPU baseline of the path is bench.py's)."""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PMCE_SYNTHETIC_BASE_DATA", "1")   # synthetic weights on the synthetic template (explicit opt-in)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=50, help="steps per timed pass")
    ap.add_argument("--min-seconds", type=float, default=5.0, help="passes are repeated until this much time has passed (at least three); the rate "
                                                                   "is the MEDIAN pass, the whole run's rate is reported as `sustained`")
    args = ap.parse_args()
    from pmce_amd import models, synth
    dev = torch.device("cuda:0")
    J, B = 17, args.batch
    sd = {k[len("pose_mesh_coevo."):]: v for k, v in synth.make_state_dict(synth.pmce_spec(J, 256, 3), seed=123).items()
          if k.startswith("pose_mesh_coevo.")}
    dec = models.CoevoDecoder.get_model(J, 256)
    dec.load_state_dict(sd)
    dec = dec.to(dev)
    dec.set_overflow_policy("report")      # asynchronous calls in the timed loop; outputs_finite is checked at the end
    g = torch.Generator().manual_seed(0)
    joints = (torch.randn(B, J, 3, generator=g) * 0.3).to(dev)                      # N(0, 0.3^2) m  (SURVEY 8d cfg 2)
    feats = torch.relu(torch.randn(B, 16, 2048, generator=g)).to(dev)
    for _ in range(5):
        dec(joints, feats)
    import statistics
    pass_s = []
    torch.cuda.synchronize(); t_all = time.perf_counter()
    while len(pass_s) < 3 or time.perf_counter() - t_all < args.min_seconds:   # (a pass of 50 steps is 50 ms: far too short to be a record alone)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            pose, mesh = dec(joints, feats)
        torch.cuda.synchronize()
        pass_s.append((time.perf_counter() - t0) / args.steps)
    total_s = time.perf_counter() - t_all
    dt = statistics.median(pass_s)
    eng = dec._ensure_packed()
    eng.set_concurrency(False); eng.profile(True)
    for _ in range(3):
        dec(joints, feats)
    torch.cuda.synchronize()
    prof = eng.profile_read(); eng.profile(False); eng.set_concurrency(True)
    # the north-star kernel against its floors: bench.py's model of it (HBM bytes, matrix instructions of the product mode in use,
    # vector-pipe floor from the committed PMC instruction count)
    import bench
    kernel_ms = {k: v[0] / 3 for k, v in prof.items() if v[1] > 0}
    launches = {k: v[1] // 3 for k, v in prof.items() if v[1] > 0}
    out = {"config": f"CoEvoDecoder-only forward, batch={B}, J=17", "clips_per_s": round(B / dt, 1), "ms_per_step": round(dt * 1e3, 4),
           "passes": len(pass_s), "steps_per_pass": args.steps, "clips_per_s_min_max": [round(B / max(pass_s), 1), round(B / min(pass_s), 1)],
           "sustained": {"clips_per_s": round(B * args.steps * len(pass_s) / total_s, 1), "seconds": round(total_s, 2)},
           "gemm_mode": eng.gemm_mode(),
           "roofline": bench.dominant_kernel_roofline({k: round(v, 4) for k, v in kernel_ms.items()}, launches, B, J, 256, eng.gemm_mode()),
           "cross_attention": bench.north_star_record(kernel_ms, launches, B, J, f16_ffn=(eng.gemm_mode() == "split_f16")),
           "kernel_ms_per_step": {k: round(v[0] / 3, 4) for k, v in prof.items() if v[1] > 0 and v[0] / 3 > 0.01},
           "outputs_finite": bool(torch.isfinite(mesh).all().item() and torch.isfinite(pose).all().item())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
