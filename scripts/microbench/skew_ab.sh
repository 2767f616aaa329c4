# bench.py quick runs: the start skew of a CU's second product workgroup (PMCE_SPLIT_SKEW; -1 = the built-in rule, 0 = none)
Q="--no-variant --no-cpu-baseline --no-latency --no-host-fed --steps 20 --warmup 5 --windows 3"
for i in 1 2 3; do for C in 512 256; do for v in -1 0 1; do
  PMCE_SPLIT_SKEW=$v timeout 300 python bench.py $Q --embed-dim $C 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('C=$C skew=$v', d['value'], d['ms_per_step'], 'lifter', d['kernel_ms_per_step'].get('gemm_lifter'))"
done; done; done
