# adaln_qkv after the next-tile prefetch: op parity, then per-kernel times from the bench's profile pass (compare with profiles/)
export PMCE_SYNTHETIC_BASE_DATA=1
timeout 300 python -m pytest tests -m gpu -x -q -k "vertex or coevo or decoder or e2e" 2>&1 | tail -3
for c in 512 256; do
timeout 200 python bench.py --embed-dim $c --steps 10 --windows 3 --no-cpu-baseline --no-host-fed --no-latency --no-variant 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernel_ms_per_step']; l=d['launches_per_step']
print(d['value'],'clips/s', d['ms_per_step'],'ms/step;', {n:(k[n], l[n]) for n in k if n in ('vertex_ca_mlp','adaln_mlp','adaln_qkv','vertex_sa','ca_fold')})"
done
