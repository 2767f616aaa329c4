for args in "--embed-dim 512" "--joints 19" "--batch 64" "--batch 1024"; do
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline $args 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('$args', '->', d['value'],'clips/s', d['ms_per_step'],'ms/step; gemm', r['achieved'],'TF; ref-equiv', d['ref_equiv_tflops'],'TF; ca', d['roofline_cross_attention']['achieved'],'GB/s')"
done
