"""Worker of the multi-rank GPU tests (tests/test_gpu_multirank.py): one process per rank; the ranks share cuda:0 on a
1-GPU box, so the collectives run over gloo there (RCCL refuses two ranks on one device) - the code path above the
backend (pmce_amd.sharding, Evaluator.evaluate) is the one the 8-GPU run uses.  mode 'nccl1' is a single rank on the RCCL
backend: every collective helper of pmce_amd.sharding on device tensors."""
import json
import os
import os.path as osp
import sys

import numpy as np
import torch

sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
os.environ.setdefault("PMCE_SYNTHETIC_BASE_DATA", "1")


def eval_inputs(N=40, seed=3):
    """Synthetic predicted / ground-truth meshes (metres) of N clips in 3 sequences; the rank boundary of a 2-way split
    (clip 20) falls INSIDE the second sequence, so its acceleration error needs joints from both ranks."""
    rng = np.random.default_rng(seed)
    base = rng.standard_normal((1, 6890, 3)).astype(np.float32) * 0.3
    drift = np.cumsum(rng.standard_normal((N, 1, 3)).astype(np.float32) * 0.01, 0)
    gt = base + drift + rng.standard_normal((N, 6890, 3)).astype(np.float32) * 0.002
    pred = gt + rng.standard_normal((N, 6890, 3)).astype(np.float32) * 0.01
    seq = np.array([0] * 12 + [1] * 17 + [2] * 11)
    return pred, gt, seq


def main():
    mode = sys.argv[1]
    from pmce_amd import sharding
    from pmce_amd.eval import Evaluator, RunningEval
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    if mode == "nccl1":
        import torch.distributed as dist
        os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1")
        # the same call pmce_amd.sharding.init_from_env makes for world > 1 (backend nccl = RCCL, communicator bound to the device)
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
        tot = sharding.reduce_metric_sums(torch.tensor([1.5, 2.0], dtype=torch.float64, device=dev))
        rows = sharding.gather_rows(torch.arange(12, dtype=torch.float32, device=dev).reshape(4, 3))
        tmax = sharding.reduce_max(0.25, dev)
        sharding.barrier()
        print(json.dumps({"backend": dist.get_backend(), "tot": tot.tolist(), "rows": rows.cpu().tolist(), "tmax": tmax,
                          "rows_device": str(rows.device)}), flush=True)
        dist.destroy_process_group()
        return
    rank, _, world = sharding.init_from_env(backend="gloo")
    pred, gt, seq = eval_inputs()
    N = len(seq)
    lo, hi = sharding.shard_range(N, rank, world)
    ev = Evaluator(dev)
    t = lambda a: torch.from_numpy(a).to(dev)
    res = ev.evaluate(t(pred[lo:hi]), t(gt[lo:hi]), seq, lo, hi)
    run = RunningEval(ev)
    for a in range(lo, hi, 7):                                  # ragged batches
        b = min(hi, a + 7)
        run.add(t(pred[a:b]), t(gt[a:b]))
    res2 = run.finish(seq, lo, hi)
    if rank == 0:
        print(json.dumps({"world": world, "evaluate": res, "running": res2}), flush=True)
    import torch.distributed as dist
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
