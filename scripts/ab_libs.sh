#!/bin/bash
# A/B/A/B of library builds inside ONE gpurun call: per-kernel-class times of a forward (scripts/microbench/ab_kernels.py).
#   bash scripts/ab_libs.sh "<tag> <tag> ..." "<batch sizes>" [C ...]      tag "new" = pmce_amd/libpmce_hip.so, others = pmce_amd/variants/
set -u
tags=${1:-"r4 new"}; batches=${2:-"1 256"}; shift 2 || true
widths=${@:-256}
export PMCE_SYNTHETIC_BASE_DATA=1
O=gpurun_out/${PMCE_ROUND:-r05}; mkdir -p $O
for C in $widths; do
  for rep in 1 2; do
    for t in $tags; do
      if [[ $t == new ]]; then unset PMCE_LIB_PATH; else export PMCE_LIB_PATH=$PWD/pmce_amd/variants/libpmce_hip_$t.so; fi
      AB_C=$C timeout 300 python scripts/microbench/ab_kernels.py $batches 2>>$O/ab.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
keys=('gemm_lifter','ln_chain','seq_attention','embed_tokens','lifter_head','gru_step','gemm_gru_in','joint_embed','ca_fold','vertex_ca_mlp','adaln_qkv','vertex_sa','adaln_mlp','tokens_kv','joint_stream','gemm_ada','gemm_final','wall_us','sum_us')
for b,v in d.items():
    if not b.startswith('B'): continue
    print('$t', 'C=$C', b, ' '.join(f'{k}={v[k]}' for k in keys if k in v))
" | tee -a $O/ab_libs.txt
    done
  done
done
unset PMCE_LIB_PATH
