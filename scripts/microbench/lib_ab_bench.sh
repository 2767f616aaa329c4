# bench.py (quick form) under each variant library, alternating:  bash scripts/microbench/lib_ab_bench.sh "<tag> <tag>" [bench args...]
export PMCE_SYNTHETIC_BASE_DATA=1
tags=$1; shift
Q="--no-variant --no-cpu-baseline --no-latency --no-host-fed --steps 20 --warmup 5 --windows 3 --sustained-seconds 0"
for rep in 1 2 3; do for t in $tags; do
  if [[ $t == new ]]; then unset PMCE_LIB_PATH; else export PMCE_LIB_PATH=$PWD/pmce_amd/variants/libpmce_hip_$t.so; fi
  timeout 200 python bench.py $Q --detail-file /tmp/d.json "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$t', '$*', d['value'], d['ms_per_step'])"
done; done
unset PMCE_LIB_PATH
