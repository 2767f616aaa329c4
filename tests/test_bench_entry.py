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
    assert r.stdout.count("\n") == 1 and r.stdout.startswith("{")      # and NOTHING else on stdout (gloo's own "[Gloo] Rank 0 is connected" goes to stderr)
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


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", osp.join(REPO, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def test_the_printed_line_is_compact_and_complete(tmp_path, capsys):
    """BENCH_r05.json came back `parsed: null`: the one stdout line had grown to 24.7 KB of nested child records.  The line bench.py prints
    is now a compact record (hard bound 4 KB) - the contract's fields + roofline + cpu_baseline - and everything else goes to the detail
    file.  Checked on the full record of a real run (committed), inflated to an 8-rank job with every optional record present."""
    bench = _bench_module()
    full = json.load(open(osp.join(REPO, "profiles", "r05_f_bench_B256.json")))
    assert len(json.dumps(full)) > 20000                                   # the record that did not parse
    full["n_gpus"] = 8
    for k in ("per_rank_clips_s", "per_rank_ms_per_step", "per_rank_sustained_clock_ghz", "per_rank_weights_load_s", "per_rank_first_step_s"):
        full[k] = full[k] * 8
    full["cpu_baseline"]["sample"] = full["cpu_baseline"]["sample"] * 20     # however wordy a nested record gets
    detail = tmp_path / "detail.json"
    bench.emit(full, str(detail))
    out = capsys.readouterr().out
    assert out.count("\n") == 1 and out.endswith("\n")                      # ONE line on stdout, the last one
    assert len(out) <= bench.COMPACT_LIMIT < 8192
    d = json.loads(out)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert k in d, k
    assert d["config"]["workload"] and d["config"]["embed_dim"] == 512 and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "launches_per_step", "source"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["source"]["achieved"].startswith("HIP events") and r["source"]["traffic"].startswith("committed profile")
    c = d["cpu_baseline"]
    assert c["value"] > 0 and c["cores"] >= 1 and c["kind"] in ("port", "reference") and len(c["sample"]) <= 160
    assert len(d["per_rank_clips_s"]) == 8 and d["detail_file"] == "detail.json"
    assert json.load(open(detail)) == full                                  # nothing is lost: the full record is in the detail file


def test_roofline_frac_is_the_algorithmic_fraction():
    """`roofline.frac` of the split-f16 product kernel = 2MNK of the fp32 products the reference asks for / time / the dense f16 peak (SURVEY
    §8d's algorithmic work); the three-times-larger rate of the f16 products the kernel ISSUES is carried as `frac_issued`."""
    bench = _bench_module()
    B, J, C = 256, 17, 512
    kernel_ms = {"gemm_lifter": 5.85, "gemm_gru_in": 0.52, "gemm_ada": 0.04, "gemm_final": 0.17, "ln_chain": 0.7}
    launches = {"gemm_lifter": 25, "gemm_gru_in": 3, "gemm_ada": 1, "gemm_final": 1, "ln_chain": 13}
    r = bench.dominant_kernel_roofline(kernel_ms, launches, B, J, C, "split_f16", clk_ghz=1.78)
    work = sum(bench.class_work(c, B, J, C)[0] for c in bench.GEMM_CLASSES)
    secs = (5.85 + 0.52 + 0.04 + 0.17) * 1e-3
    assert r["kernel"] == "gemm_split_kernel" and r["bound"] == "mfma" and r["peak"] == 2500.0
    assert abs(r["achieved"] - work / secs / 1e12) < 0.1 and abs(r["frac"] - work / secs / 1e12 / 2500.0) < 1e-3
    assert abs(r["frac_issued"] - 3 * r["frac"]) < 2e-3 and abs(r["algorithmic_per_launch"] - work / 30) < 1.0


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


def test_rank_cpu_slices_and_shared_weights(monkeypatch, tmp_path):
    """Multi-GPU polish of round 4, host logic only: ranks bind to disjoint contiguous CPU slices; the synthetic
    weights are generated by local rank 0 alone and mapped by the others; the streaming flag of the roofline arithmetic removes the
    per-frame products from a window batch's work."""
    import os
    import torch
    sys.path.insert(0, REPO)
    import bench
    from pmce_amd import synth
    if hasattr(os, "sched_getaffinity"):
        before = os.sched_getaffinity(0)
        try:
            if len(before) >= 2:
                a = bench.bind_rank_to_cpus(0, 2)
                os.sched_setaffinity(0, before)
                b = bench.bind_rank_to_cpus(1, 2)
                assert a and b and not (set(a) & set(b)) and set(a) | set(b) <= set(before)
                assert os.sched_getaffinity(0) == set(b)
            assert bench.bind_rank_to_cpus(0, 1) is None
        finally:
            os.sched_setaffinity(0, before)
    calls = []
    small = {"w": torch.arange(12.0).reshape(3, 4), "b": torch.ones(5)}
    monkeypatch.setattr(synth, "make_state_dict", lambda spec, seed=123: (calls.append(1), {k: v.clone() for k, v in small.items()})[1])
    monkeypatch.setenv("MASTER_PORT", "45678")
    sd0, path = bench.shared_state_dict(17, 256, 0, lambda: None)
    try:
        assert path and os.path.exists(path) and len(calls) == 1
        sd1, _ = bench.shared_state_dict(17, 256, 1, lambda: None)            # a non-zero local rank maps the file: no second generation
        assert len(calls) == 1 and all(torch.equal(sd1[k], small[k]) for k in small)
    finally:
        if path and os.path.exists(path):
            os.remove(path)
    full, _ = bench.class_work("gemm_lifter", 256, 17, 512)
    win, _ = bench.class_work("gemm_lifter", 256, 17, 512, streaming=True)
    assert 0.75 * full < win < full                                            # 5 of 6 blocks, no imgfeat_embed
    assert bench.class_work("gemm_gru_in", 256, 17, 512, streaming=True)[0] < 0.4 * bench.class_work("gemm_gru_in", 256, 17, 512)[0]   # layer 1 only
