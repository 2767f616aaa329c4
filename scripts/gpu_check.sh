#!/bin/bash
# One gpurun call: GPU parity tests, a short bench and a rocprofv3 kernel-trace of the bench command.
# Usage (from the repo root, on the GPU box):  bash scripts/gpu_check.sh [tests|bench|prof|all]
set -u
what=${1:-all}
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import pmce_amd.build as b; print(b.build())" > gpurun_out/build.log 2>&1
if [[ $what == all || $what == tests ]]; then
  timeout 1200 python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
  tail -n 60 gpurun_out/pytest_gpu.log
fi
if [[ $what == all || $what == bench ]]; then
  timeout 600 python bench.py --steps 10 --warmup 2 > gpurun_out/bench.log 2> gpurun_out/bench.err
  echo "bench exit: $?"; cat gpurun_out/bench.log; tail -n 5 gpurun_out/bench.err
fi
if [[ $what == all || $what == prof ]]; then
  rm -rf gpurun_out/prof
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/prof -o trace -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --single-stream > $OLDPWD/gpurun_out/prof.log 2>&1)
  echo "rocprof exit: $?"
  find gpurun_out/prof -name "*stats*" | head
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
  [[ -n "$f" ]] && head -n 30 "$f"
  # keep only the small summaries
  find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
fi
