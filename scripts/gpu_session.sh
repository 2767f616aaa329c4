#!/bin/bash
# GPU session script: bash scripts/gpu_session.sh [tests] [bench] [order_ab] [prof] [pmc] [quick "<pytest -k expr>"] (any subset, in this order).
# Everything lands under gpurun_out/r04/.
set -u
R=${PMCE_ROUND:-r04}
mkdir -p gpurun_out/$R
O=gpurun_out/$R
export TMPDIR=/tmp PMCE_SYNTHETIC_BASE_DATA=1
python -c "import pmce_amd.build as b; print(b.build()); print(b.build_diag())" > $O/build.log 2>&1 || { cat $O/build.log; exit 1; }
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
  timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err
  echo "bench exit: $?"; python scripts/show_bench.py $O/bench.json 2>/dev/null | head -120 || head -c 3000 $O/bench.json
  tail -n 5 $O/bench.err | grep -v amdgpu.ids || true
  ;;
order_ab)
  for i in 1 2; do for o in 0 1; do
    PMCE_SPLIT_ORDER=$o timeout 300 python bench.py $QUICK > $O/bench_order${o}_$i.json 2>> $O/order_ab.err
    python - <<PY
import json
d=json.loads(open("$O/bench_order${o}_$i.json").read().strip().splitlines()[-1])
print("order $o run $i: ms/step", d["ms_per_step"], "gemm_lifter", d["kernel_ms_per_step"].get("gemm_lifter"), "gemm_gru_in", d["kernel_ms_per_step"].get("gemm_gru_in"), "clock", d["roofline"].get("sustained_clock_ghz"))
PY
  done; done
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
    python scripts/pmc_summary.py $O/pmc$C $O/pmc_hbm_traffic_per_launch_C$C.json
    find $O/pmc$C -name "*.csv" -size +8M -delete
  done
  ;;
esac
done
exit 0
