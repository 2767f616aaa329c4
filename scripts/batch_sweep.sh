for b in 32 64 128 256 512; do python bench.py --steps 10 --warmup 2 --batch $b --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); k=d['kernel_ms_per_step']
print('B=%d'%$b, d['value'],'clips/s', d['ms_per_step'],'ms', 'gemm_lifter',k['gemm_lifter'],'attn',k['seq_attention'],'ln',k['ln_chain'],'gru_in',k['gemm_gru_in'],'gru_step',k['gru_step'],'mlp',k['adaln_mlp'],'sa',k['vertex_sa'])"; done
