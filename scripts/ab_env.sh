#!/bin/bash
# A/B/A/B of ONE library under two (or more) values of an environment switch inside ONE gpurun call: per-kernel-class times of a forward
# (scripts/microbench/ab_kernels.py) - e.g.  bash scripts/ab_env.sh PMCE_LN_FOLD "1 0" "256" 512
set -u
var=$1; values=$2; batches=${3:-"256"}; shift 3 || true
widths=${@:-512}
export PMCE_SYNTHETIC_BASE_DATA=1
O=gpurun_out/${PMCE_ROUND:-r06}; mkdir -p $O
for C in $widths; do
  for rep in 1 2 3; do
    for v in $values; do
      export $var=$v
      AB_C=$C timeout 300 python scripts/microbench/ab_kernels.py $batches 2>>$O/ab.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
keys=('gemm_lifter','ln_chain','seq_attention','gru_step','gemm_gru_in','wall_us','sum_us')
for b,x in d.items():
    if not b.startswith('B'): continue
    print('$var=$v', 'C=$C', b, ' '.join(f'{k}={x[k]}' for k in keys if k in x))
" | tee -a $O/ab_env.txt
    done
  done
done
unset $var
