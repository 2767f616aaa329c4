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


def test_roofline_records_of_the_non_gemm_kernels():
    """bench.py's records for the HBM-bound attention kernel and the north-star cross-attention kernel: pure functions of the measured
    kernel times and of the committed PMC summaries (no GPU)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", osp.join(REPO, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    B, J, C = 256, 17, 512
    kernel_ms = {"seq_attention": 0.72, "vertex_ca_mlp": 0.264}
    launches = {"seq_attention": 6, "vertex_ca_mlp": 3}
    a = bench.attention_record(kernel_ms, launches, B, J, C, True)
    assert a["kernel"] == "seq_attention_mfma_kernel" and a["bound"] == "hbm" and a["launches_per_step"] == 6
    assert a["algorithmic_bytes_per_launch"] == B * 16 * J * 4 * C * 4                     # q, k, v read + result written, once each
    assert abs(a["achieved"] - a["algorithmic_bytes_per_launch"] / 0.12e-3 / 1e9) < 1.0 and abs(a["frac"] - a["achieved"] / 8000.0) < 1e-3
    if a["traffic"] is not None:                                                           # the committed PMC pass: no wasted re-reads
        assert 0.98 < a["traffic"] / a["algorithmic_bytes_per_launch"] < 1.05
    assert bench.attention_record(kernel_ms, launches, B, J, C, False)["kernel"] == "seq_attention_pair_kernel"
    assert bench.attention_record({}, {}, B, J, C, True) is None
    n = bench.north_star_record(kernel_ms, launches, B, J, f16_ffn=True)
    assert n["kernel"] == "vertex_ca_mlp" and n["bytes_per_clip_dir_block"] == 229376
    assert abs(n["hbm_floor_ms"] - 229376.0 * B / 8e12 * 1e3) < 1e-4 and n["mfma_floor_ms"] > n["hbm_floor_ms"]
    if "valu_floor_ms" in n:                                                               # from the committed instruction count
        assert abs(n["serial_floor_ms"] - (n["mfma_floor_ms"] + n["valu_floor_ms"])) < 1e-4
        assert 0.0 < n["frac_of_serial_floor"] < 1.0 and n["bound"] in ("valu", "mfma")
