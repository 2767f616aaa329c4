"""The N>1 entry point of bench.py without a GPU: `python bench.py --gpus N` must spawn its own N ranks when
no launcher set WORLD_SIZE (the form the round driver uses for the scaling curve), and must also run under
torch.distributed.run.  `--dist-check` stops after the rendezvous and the path's three collectives (gloo here, RCCL on GPUs)."""
import json
import os
import os.path as osp
import subprocess
import sys

REPO = osp.dirname(osp.dirname(osp.abspath(__file__)))


def _run(cmd, extra_env=None, timeout=300):
    env = dict(os.environ, PMCE_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    r = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, [json.loads(l) for l in lines]


def test_bench_spawns_its_own_ranks():
    r, lines = _run([sys.executable, "bench.py", "--gpus", "2", "--dist-check"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout                       # rank 0 only
    d = lines[0]
    assert d["n_gpus"] == 2 and d["ranks"] == 2 and d["backend"] == "gloo"
    assert d["clips"] == 1000 and d["gathered"] == 1000 and d["max_rank"] == 1


def test_bench_under_torch_distributed_run():
    r, lines = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                     "127.0.0.1", "--master-port", "29541", "bench.py", "--gpus", "2", "--dist-check"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["gathered"] == 1000


def test_bench_single_rank_and_mismatch():
    r, lines = _run([sys.executable, "bench.py", "--dist-check"])
    assert r.returncode == 0 and lines[0]["n_gpus"] == 1 and lines[0]["ranks"] == 1
    # a launcher that set WORLD_SIZE=1 while --gpus 2 was asked for is an error, not a silent 1-GPU run
    r, _ = _run([sys.executable, "bench.py", "--gpus", "2", "--dist-check"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_failed_rank_takes_the_job_down():
    """A rank that dies (here: rank 1 cannot find a device in a GPU-less container) ends the whole job with a non-zero
    exit code instead of leaving rank 0 waiting in a barrier."""
    r, lines = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "1", "--windows", "1"], timeout=600)
    assert r.returncode != 0 and not lines
