# vertex-stream CrossAttentionBlock: fused (one launch) vs vertex_ca + adaln_mlp (two), same box
export PMCE_SYNTHETIC_BASE_DATA=1
for f in 1 0; do
  echo "== PMCE_VERTEX_FUSED=$f"
  PMCE_VERTEX_FUSED=$f python bench.py --embed-dim 256 --steps 10 --windows 3 --no-cpu-baseline --no-host-fed --no-latency --no-variant 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernel_ms_per_step']; l=d['launches_per_step']
print(d['value'],'clips/s', d['ms_per_step'],'ms/step; decoder kernels:', {n:(k[n], l[n]) for n in k if n in ('vertex_ca','vertex_ca_mlp','adaln_mlp','adaln_qkv','vertex_sa','ca_fold')})
print('   north-star:', d['roofline_cross_attention'])"
done
