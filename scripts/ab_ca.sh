python -c "import pmce_amd.build as b; b.build(force=True)" > /dev/null 2>&1
python -m pytest tests/test_gpu_ops.py -m gpu -q -k "cross_attn" -p no:cacheprovider 2>&1 | tail -2
echo "== launch_bounds(448,4): 128 VGPR + 20B scratch"
for b in 256 1024; do python bench.py --steps 5 --warmup 2 --batch $b --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('B=$b', d['value'], 'vertex_ca ms', d['kernel_ms_per_step']['vertex_ca'], d['roofline_cross_attention'])"; done
sed -i 's/__global__ __launch_bounds__(448, 4) void vertex_ca_kernel(/__global__ __launch_bounds__(448) void vertex_ca_kernel(/' pmce_amd/csrc/coevo.hip
python -c "import pmce_amd.build as b; b.build(force=True)" > /dev/null 2>&1
echo "== launch_bounds(448): 130 VGPR"
for b in 256 1024; do python bench.py --steps 5 --warmup 2 --batch $b --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('B=$b', d['value'], 'vertex_ca ms', d['kernel_ms_per_step']['vertex_ca'], d['roofline_cross_attention'])"; done
