#!/bin/bash
# Other configurations of `bench.py` (one line each): the 3DPW joint count, other batch sizes, both widths.
for args in "--joints 19" "--batch 64" "--batch 1024" "--embed-dim 256 --joints 19" "--embed-dim 256 --batch 64" "--embed-dim 256 --batch 1024" "--pipeline-depth 1" "--pipeline-depth 3"; do
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-latency --no-variant --no-host-fed $args 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']; a=d.get('roofline_attention') or {}
print('%-34s' % '$args', '->', d['value'],'clips/s', d['ms_per_step'],'ms/step; product kernel', r['achieved'], r['unit'], '(%.3f)' % r['frac'], '; ref-equiv', d['ref_equiv_tflops'],'TFLOP/s; attention', a.get('kernel'), a.get('achieved'),'GB/s')"
done
