#!/bin/bash
# GPU session script: bash scripts/gpu_session.sh [tests] [bench] [sweep] [prof] [pmc] (any subset, in this order).
# Everything lands under gpurun_out/r03/.
set -u
mkdir -p gpurun_out/r03
O=gpurun_out/r03
export TMPDIR=/tmp PMCE_SYNTHETIC_BASE_DATA=1
python -c "import pmce_amd.build as b; print(b.build())" > $O/build.log 2>&1 || { cat $O/build.log; exit 1; }
for what in "$@"; do
case $what in
tests)
  timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $O/pytest_gpu.log 2>&1
  echo "pytest exit: $?" | tee -a $O/pytest_gpu.log
  grep -E "passed|failed|error" $O/pytest_gpu.log | tail -5
  ;;
bench)
  timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
  echo "bench exit: $?"; python scripts/show_bench.py $O/bench.json 2>/dev/null | head -80 || head -c 3000 $O/bench.json
  tail -n 5 $O/bench.err | grep -v amdgpu.ids || true
  ;;
sweep)
  bash scripts/sweep_r02.sh > $O/sweep.log 2>&1; cp gpurun_out/sweep_r02.txt $O/ 2>/dev/null; cat $O/sweep_r02.txt
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
