"""Multi-rank paths on the GPU box: clips shard contiguously over ranks, the only collective is the final
metric reduction.  (1) `Evaluator.evaluate` / `RunningEval` over a 2-rank split whose boundary cuts a sequence == the
unsharded result == the metrics oracle; (2) every collective helper on the RCCL backend; (3) `python bench.py --gpus 2`
spawns two ranks that run the timed loop (sharing the one GPU of this box) and print ONE line with n_gpus = 2."""
import json
import os
import os.path as osp
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = osp.dirname(osp.abspath(__file__))
REPO = osp.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spawn(mode, world):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, osp.join(HERE, "_rank_worker.py"), mode], env=env, cwd=REPO,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-3000:]
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    return json.loads(lines[-1])


def test_sharded_evaluate_across_a_sequence_boundary():
    sys.path.insert(0, HERE)
    from _rank_worker import eval_inputs
    from oracle import metrics_oracle as MO
    from pmce_amd import assets
    from pmce_amd.eval import Evaluator
    pred, gt, seq = eval_inputs()
    dev = torch.device("cuda:0")
    ev = Evaluator(dev)
    one = ev.evaluate(torch.from_numpy(pred).to(dev), torch.from_numpy(gt).to(dev), seq)
    two = _spawn("eval", 2)
    assert two["world"] == 2
    jr = assets.load_j_regressor("h36m").astype(np.float32)
    ref = MO.evaluate_samples(pred.astype(np.float64) * 1000, gt.astype(np.float64) * 1000, jr[:1], 0, jr, seq)
    print("unsharded", one, "\n2 ranks  ", two["evaluate"], "\noracle   ", {k: ref[k] for k in ("MPVPE", "MPJPE", "PA_MPJPE", "ACCEL")})
    for got in (two["evaluate"], two["running"]):
        for k in ("MPVPE", "MPJPE", "PA-MPJPE", "ACCEL"):
            assert abs(got[k] - one[k]) < 1e-4, (k, got[k], one[k])           # mm; sharded == unsharded
        assert got["samples"] == one["samples"] == len(seq)
    assert abs(one["MPVPE"] - ref["MPVPE"]) < 1e-3 and abs(one["MPJPE"] - ref["MPJPE"]) < 1e-3
    assert abs(one["PA-MPJPE"] - ref["PA_MPJPE"]) < 1e-3 and abs(one["ACCEL"] - ref["ACCEL"]) < 1e-3


def test_collectives_on_rccl():
    d = _spawn("nccl1", 1)
    assert d["backend"] == "nccl" and d["tot"] == [1.5, 2.0] and d["tmax"] == 0.25
    assert d["rows"] == np.arange(12, dtype=np.float32).reshape(4, 3).tolist() and d["rows_device"].startswith("cuda")


def test_bench_two_ranks_on_this_box():
    env = dict(os.environ, PMCE_BENCH_SHARE_GPU="1", PMCE_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--windows", "2", "--batch",
                        "32", "--embed-dim", "256", "--cpu-seconds", "2", "--sustained-seconds", "1", "--detail-file", "/tmp/pmce_bench_detail_2r.json"],
                       cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout.count("\n") == 1 and len(r.stdout) < 8192, r.stdout[:2000]     # ONE compact line is all of stdout
    d = json.loads(r.stdout)
    print({k: d[k] for k in ("value", "n_gpus", "ms_per_step", "outputs_finite")})
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 64 and d["outputs_finite"] and d["value"] > 0
    assert d["roofline"] is not None and d["cpu_baseline"]["value"] > 0           # rank 0 times the CPU baseline at N > 1 as well
    assert len(d["per_rank_clips_s"]) == 2 and all(v > 0 for v in d["per_rank_clips_s"])
    assert d["metric_reduction"]["clips_counted"] == 64 and d["metric_reduction"]["backend"] == "gloo"


def test_bench_two_ranks_over_rccl_when_two_gpus_are_visible():
    """The first box with two devices exercises RCCL with more than one rank: `python bench.py --gpus 2` exactly as the round
    driver launches it (backend nccl, one rank per GPU).  On a one-GPU box RCCL refuses two ranks on one device - skipped."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs (RCCL refuses two ranks on one device)")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                            "PMCE_BENCH_SHARE_GPU", "PMCE_DIST_BACKEND")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "5", "--warmup", "2", "--windows", "2", "--cpu-seconds", "2"],
                       cwd=REPO, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout.count("\n") == 1 and len(r.stdout) < 8192
    d = json.loads(r.stdout)
    print({k: d[k] for k in ("value", "n_gpus", "ms_per_step", "per_rank_clips_s", "metric_reduction")})
    assert d["n_gpus"] == 2 and d["metric_reduction"]["backend"] == "nccl" and d["metric_reduction"]["clips_counted"] == 512
    assert d["outputs_finite"] and min(d["per_rank_clips_s"]) > 0.5 * max(d["per_rank_clips_s"])


def test_bench_eight_ranks_dry_run_prints_one_compact_line():
    """Dry run of the line the driver's 8-GPU scaling run will parse: eight ranks share this box's one device (gloo for the
    metric reduction), tiny batch.  Asserted: stdout is ONE line under 8 KB that parses, n_gpus 8, eight entries per per-rank list, roofline
    and cpu_baseline present; the detail file holds the full record.  No scaling number is read off this."""
    env = dict(os.environ, PMCE_BENCH_SHARE_GPU="1", PMCE_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    detail = "/tmp/pmce_bench_detail_8r.json"
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--steps", "2", "--warmup", "1", "--windows", "1", "--batch", "8",
                        "--embed-dim", "256", "--cpu-seconds", "1", "--sustained-seconds", "1", "--detail-file", detail],
                       cwd=REPO, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout.count("\n") == 1 and len(r.stdout) < 8192, r.stdout[:2000]
    d = json.loads(r.stdout)
    assert d["n_gpus"] == 8 and d["config"]["global_batch"] == 64 and d["outputs_finite"] and d["value"] > 0
    assert d["roofline"]["frac"] > 0 and d["cpu_baseline"]["value"] > 0 and d["config"]["workload"] and d["dtype"] == "f32"
    assert len(d["per_rank_clips_s"]) == 8 and all(v > 0 for v in d["per_rank_clips_s"])
    assert d["metric_reduction"] == {"clips_counted": 64, "backend": "gloo"}
    full = json.load(open(detail))
    for k in ("per_rank_clips_s", "per_rank_ms_per_step", "per_rank_sustained_clock_ghz", "per_rank_weights_load_s", "per_rank_first_step_s"):
        assert len(full[k]) == 8, k
    assert full["value"] == d["value"] and full["kernel_ms_per_step"]
