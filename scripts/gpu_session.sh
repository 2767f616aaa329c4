#!/bin/bash
# GPU session script: bash scripts/gpu_session.sh [tests] [prof] [pmc] [counters] [install] [bench] [ab_r4] [quick "<pytest -k expr>"] (any subset, any order;
# `install` after pmc / counters and before bench makes the bench line read the PMC summaries of its own build).
# Everything lands under gpurun_out/$PMCE_ROUND (default r05); summaries are copied to profiles/ by hand afterwards.
set -u
R=${PMCE_ROUND:-r06}
mkdir -p gpurun_out/$R
O=gpurun_out/$R
export TMPDIR=/tmp PMCE_SYNTHETIC_BASE_DATA=1
python -c "import pmce_amd.build as b; print(b.build()); print(b.build_diag()); from pmce_amd import _lib; print('build id', _lib.build_id())" > $O/build.log 2>&1 || { cat $O/build.log; exit 1; }
tail -n 1 $O/build.log
QUICK="--no-variant --no-cpu-baseline --no-latency --no-host-fed --steps 20 --warmup 5 --windows 3"
while [[ $# -gt 0 ]]; do
what=$1; shift
case $what in
tests)
  timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $O/pytest_gpu.log 2>&1
  echo "pytest exit: $?" | tee -a $O/pytest_gpu.log
  grep -E "passed|failed|error" $O/pytest_gpu.log | tail -5
  grep -E "^(FAILED|ERROR)" $O/pytest_gpu.log | head -20
  ;;
quick)
  expr=$1; shift
  timeout 900 python -m pytest tests -m gpu -q -s -x -p no:cacheprovider -k "$expr" > $O/pytest_quick.log 2>&1
  echo "pytest quick exit: $?"; tail -n 40 $O/pytest_quick.log
  ;;
bench)
  timeout 1200 python bench.py --detail-file $O/bench_detail.json > $O/bench_line.json 2> $O/bench.err
  echo "bench exit: $?"; echo "stdout: $(wc -l < $O/bench_line.json) line(s), $(wc -c < $O/bench_line.json) bytes"; cat $O/bench_line.json
  python scripts/show_bench.py $O/bench_detail.json 2>/dev/null | head -120 || head -c 3000 $O/bench_detail.json
  tail -n 5 $O/bench.err | grep -v amdgpu.ids || true
  ;;
ab_r4)
  # the round-4 tree (git archive 4bbd38d into .r4tree, library prebuilt) against this tree, A/B/A/B on THIS box, both widths
  rm -f $O/ab_r4_vs_r5.txt
  for C in 512 256; do for i in 1 2; do for T in r4 r5; do
    if [[ $T == r4 ]]; then (cd .r4tree && timeout 300 python bench.py --embed-dim $C $QUICK 2>> ../$O/ab_r4.err) > $O/ab_tmp.json
    else timeout 300 python bench.py --embed-dim $C $QUICK --sustained-seconds 0 --detail-file $O/ab_tmp.json > /dev/null 2>> $O/ab_r4.err; fi
    T=$T C=$C python - <<PY | tee -a $O/ab_r4_vs_r5.txt
import json, os
d = json.loads(open("$O/ab_tmp.json").read().strip().splitlines()[-1])
k = d["kernel_ms_per_step"]
dec = sum(k.get(n, 0) for n in ("joint_embed", "ca_fold", "vertex_ca_mlp", "adaln_qkv", "vertex_sa", "adaln_mlp"))
print(os.environ["T"], "C=" + os.environ["C"], "clips/s", d["value"], "ms/step", d["ms_per_step"], "| vertex stream per CoevoBlock us", round(dec / 3 * 1e3, 1),
      "| gemm_lifter", k.get("gemm_lifter"), "ln_chain", k.get("ln_chain"), "seq_attention", k.get("seq_attention"), "tokens_kv", k.get("tokens_kv"),
      "| single-stream sum", d["kernel_ms_total_single_stream"], "clock", d["roofline"].get("sustained_clock_ghz"))
PY
  done; done; done
  ;;
prof)
  for C in 512 256; do
    rm -rf $O/prof$C
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof$C -o trace -- python $OLDPWD/bench.py --embed-dim $C --steps 5 --warmup 2 --windows 1 --no-cpu-baseline --no-latency --no-variant --single-stream > $OLDPWD/$O/prof$C.log 2>&1)
    echo "rocprof C=$C exit: $?"
    f=$(find $O/prof$C -name "*kernel_stats.csv" | head -1)
    [[ -n "$f" ]] && cp "$f" $O/kernel_stats_C$C.csv && head -n 16 "$f" | cut -c1-160
    find $O/prof$C -name "*kernel_trace.csv" -size +20M -delete
  done
  ;;
pmc)
  for C in 512 256; do
    for c in FETCH_SIZE WRITE_SIZE; do
      rm -rf $O/pmc$C/$c
      (cd /tmp && timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OLDPWD/$O/pmc$C/$c -o pmc -- python $OLDPWD/bench.py --embed-dim $C --steps 2 --warmup 1 --windows 1 --no-cpu-baseline --no-latency --no-variant --single-stream > $OLDPWD/$O/pmc$C.$c.log 2>&1)
      echo "pmc C=$C $c exit $?"
    done
    python scripts/pmc_summary.py $O/pmc$C $O/pmc_hbm_traffic_per_launch_C$C.json | grep -v "at::\|rocclr" | head -40
    find $O/pmc$C -name "*.csv" -size +8M -delete
  done
  ;;
install)
  # the PMC summaries of THIS session become the files bench.py reads (profiles/ on the box; the same copies are committed afterwards), so that
  # the bench line of the session carries roofline fields of its own build (stale: false)
  for C in 512 256; do
    [[ -s $O/pmc_hbm_traffic_per_launch_C$C.json ]] && cp $O/pmc_hbm_traffic_per_launch_C$C.json profiles/pmc_hbm_traffic_per_launch_C$C.json
    [[ -s $O/pmc_counters_per_kernel_C$C.json ]] && cp $O/pmc_counters_per_kernel_C$C.json profiles/pmc_counters_per_kernel_C$C.json
  done
  [[ -s $O/pmc_counters_per_kernel_C512.json ]] && cp $O/pmc_counters_per_kernel_C512.json profiles/pmc_counters_per_kernel.json
  python -c "import bench; print('bench will read:', bench._pmc_file(512), bench.library_build_id())" 2>/dev/null || true
  ;;
counters)
  # SQ counters per kernel (matrix-pipe busy, VALU instruction counts, wait cycles, LDS conflicts) of the whole forward, both widths
  for C in 512 256; do
    PMCE_PMC_C=$C bash scripts/pmc_kernels.sh "." > $O/pmc_counters_C$C.txt 2>&1
    cp gpurun_out/kpmc/summary.json $O/pmc_counters_per_kernel_C$C.json
    grep -A1 "vertex_ca_mlp\|gemm_split_kernel<2, 4, 0, true" $O/pmc_counters_C$C.txt | head -8
    grep "matrix pipe busy" $O/pmc_counters_C$C.txt | wc -l
  done
  ;;
esac
done
exit 0
