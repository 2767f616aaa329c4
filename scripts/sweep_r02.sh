#!/bin/bash
# Concurrency knobs of the whole path, one short bench each (round 2): pipeline depth / stagger / persistent GEMM
# workgroups per CU.  Output: gpurun_out/sweep_r02.txt (one line per configuration).
set -u
mkdir -p gpurun_out
out=gpurun_out/sweep_r02.txt
: > $out
run() {  # label, env..., -- args
  local label=$1; shift
  local envs=()
  while [[ $1 != -- ]]; do envs+=("$1"); shift; done
  shift
  local line
  line=$(env "${envs[@]}" timeout 300 python bench.py --steps 10 --warmup 2 --windows 3 --no-cpu-baseline --no-host-fed --no-latency --no-variant "$@" 2>/dev/null | tail -1)
  python - "$label" "$line" >> $out <<'PY'
import json, sys
label, line = sys.argv[1], sys.argv[2]
try:
    d = json.loads(line)
    print(f"{label:46s} {d['value']:9.1f} clips/s  {d['ms_per_step']:8.3f} ms/step  windows {d['windows']['ms_per_step']}")
except Exception as e:
    print(f"{label:46s} FAILED {e} {line[:200]}")
PY
}
for C in 512 256; do
  run "C=$C depth2 stagger (default)"        X=1 -- --embed-dim $C
  run "C=$C depth2 free-running"             X=1 -- --embed-dim $C --no-stagger
  run "C=$C depth1"                          X=1 -- --embed-dim $C --pipeline-depth 1
  run "C=$C depth3 free-running"             X=1 -- --embed-dim $C --pipeline-depth 3 --no-stagger
  run "C=$C gemm-grid1 depth2 free-running"  PMCE_GEMM_GRID=1 -- --embed-dim $C --no-stagger
  run "C=$C gemm-grid1 depth3 free-running"  PMCE_GEMM_GRID=1 -- --embed-dim $C --pipeline-depth 3 --no-stagger
  run "C=$C gemm-grid1 depth2 stagger"       PMCE_GEMM_GRID=1 -- --embed-dim $C
done
cat $out
