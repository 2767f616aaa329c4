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
    ap.add_argument("--steps", type=int, default=50)
    args = ap.parse_args()
    from pmce_amd import models, synth
    dev = torch.device("cuda:0")
    J, B = 17, args.batch
    sd = {k[len("pose_mesh_coevo."):]: v for k, v in synth.make_state_dict(synth.pmce_spec(J, 256, 3), seed=123).items()
          if k.startswith("pose_mesh_coevo.")}
    dec = models.CoevoDecoder.get_model(J, 256)
    dec.load_state_dict(sd)
    dec = dec.to(dev)
    g = torch.Generator().manual_seed(0)
    joints = (torch.randn(B, J, 3, generator=g) * 0.3).to(dev)                      # N(0, 0.3^2) m  (SURVEY 8d cfg 2)
    feats = torch.relu(torch.randn(B, 16, 2048, generator=g)).to(dev)
    for _ in range(5):
        dec(joints, feats)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(args.steps):
        pose, mesh = dec(joints, feats)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    eng = dec._ensure_packed()
    eng.set_concurrency(False); eng.profile(True)
    for _ in range(3):
        dec(joints, feats)
    torch.cuda.synchronize()
    prof = eng.profile_read(); eng.profile(False); eng.set_concurrency(True)
    # the north-star kernel: since round 2 the whole CrossAttentionBlock (cross-attention + FFN) is one launch
    kname = "vertex_ca_mlp" if prof.get("vertex_ca_mlp", (0, 0))[1] > 0 else "vertex_ca"
    ca_ms = prof[kname][0] / prof[kname][1]
    mfma_per_tile = 64 + 16 * ((J + 7) // 8) + (512 if kname == "vertex_ca_mlp" else 0)
    mfma_floor_ms = B * 14 * mfma_per_tile * 64 / 1024 / 2.4e9 * 1e3
    hbm_floor_ms = 229376.0 * B / 8e12 * 1e3
    out = {"config": f"CoEvoDecoder-only forward, batch={B}, J=17", "clips_per_s": round(B / dt, 1), "ms_per_step": round(dt * 1e3, 4),
           "cross_attention": {"kernel": kname, "avg_launch_ms": round(ca_ms, 5), "bytes_per_clip_dir_block": 229376,
                               "achieved_GBps": round(229376 * B / (ca_ms * 1e-3) / 1e9, 1), "peak_GBps": 8000,
                               "mfma_floor_ms": round(mfma_floor_ms, 5), "hbm_floor_ms": round(hbm_floor_ms, 5),
                               "frac_of_floor": round(max(mfma_floor_ms, hbm_floor_ms) / ca_ms, 4)},
           "kernel_ms_per_step": {k: round(v[0] / 3, 4) for k, v in prof.items() if v[1] > 0 and v[0] / 3 > 0.01},
           "outputs_finite": bool(torch.isfinite(mesh).all().item() and torch.isfinite(pose).all().item())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
