export PMCE_SYNTHETIC_BASE_DATA=1
for C in 256 512; do python bench.py --embed-dim $C --steps 20 --windows 5 --no-cpu-baseline --no-host-fed --no-latency --no-variant 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print(d['config']['embed_dim'], d['value'], d['windows']['ms_per_step'], 'gemm_lifter', d['kernel_ms_per_step']['gemm_lifter'], 'frac', d['roofline']['frac'])"; done
