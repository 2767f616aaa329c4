#!/usr/bin/env python3
"""Soak run of the HIP path (GPU only): forwards of changing batch sizes through the two pipeline lanes and through direct calls for --seconds,
every result compared BIT FOR BIT with the first result of the same batch (the path is deterministic: same kernels, same order, no atomics in
the arithmetic), the overflow word and finiteness checked throughout.  Prints one JSON line."""
import argparse, json, os, random, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PMCE_SYNTHETIC_BASE_DATA", "1")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--embed-dim", type=int, default=512)
    ap.add_argument("--joints", type=int, default=17)
    ap.add_argument("--batches", default="1,2,3,8,16,31,64,100,128,192,255,256")
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    from pmce_amd import assets, models, synth
    dev = torch.device("cuda:0")
    J, C = args.joints, args.embed_dim
    model = models.PMCE.get_model(J, C, 3)
    model.load_state_dict(synth.make_state_dict(synth.pmce_spec(J, C, 3), seed=123))
    model.set_j_regressor(assets.load_j_regressor("h36m"))
    model = model.to(dev)
    model.set_overflow_policy("report")
    sizes = [int(x) for x in args.batches.split(",")]
    p_all, f_all = synth.make_inputs(max(sizes), J, 11)
    p_all, f_all = torch.from_numpy(p_all).to(dev), torch.from_numpy(f_all).to(dev)
    pipe = model.pipeline(2)
    rng = random.Random(args.seed)
    first, counts, mismatches, nonfinite = {}, {}, [], 0
    t0 = time.perf_counter()
    it = 0
    while time.perf_counter() - t0 < args.seconds:
        burst = [rng.choice(sizes) for _ in range(rng.randint(1, 6))]
        direct = rng.random() < 0.3
        outs = []
        for B in burst:
            # a different window of the input pool per batch size would change the expected result: always the first B clips
            if direct:
                outs.append((B, model.forward_with_joints(p_all[:B], f_all[:B])))
            else:
                outs.append((B, pipe.submit(p_all[:B], f_all[:B])))
        for B, o in outs:
            res = o if direct else o.result()
            res = [t for t in res if t is not None]
            torch.cuda.current_stream().synchronize()
            if not all(bool(torch.isfinite(t).all()) for t in res):
                nonfinite += 1
            key = B
            if key not in first:
                first[key] = [t.clone() for t in res]
            elif not all(torch.equal(a, b) for a, b in zip(first[key], res)):
                mismatches.append({"iteration": it, "batch": B, "direct": direct,
                                   "max_abs": max(float((a - b).abs().max()) for a, b in zip(first[key], res))})
            counts[B] = counts.get(B, 0) + 1
            it += 1
    pipe.synchronize()
    # a batch's clips do not depend on the batch they rode in: the first clip of every size against the B = 1 result
    cross = None
    if 1 in first:
        cross = max(float((first[B][0][:1] - first[1][0]).abs().max()) for B in first)
    print(json.dumps({"seconds": round(time.perf_counter() - t0, 1), "forwards": it, "clips": sum(b * n for b, n in counts.items()),
                      "per_batch": dict(sorted(counts.items())), "bit_mismatches": len(mismatches), "first_mismatches": mismatches[:5],
                      "nonfinite_results": nonfinite, "overflow_word": bool(model.overflowed()),
                      "mesh_of_clip_0_max_abs_across_batch_sizes_m": cross, "embed_dim": C, "joints": J}))
    return 1 if mismatches or nonfinite else 0


if __name__ == "__main__":
    sys.exit(main())
