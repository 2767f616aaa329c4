python -m pytest tests/test_gpu_e2e.py -m gpu -q -k "reference_fixture or odd_batch" -p no:cacheprovider 2>&1 | tail -2
PMCE_GRU_NQ=2 python -m pytest tests/test_gpu_e2e.py -m gpu -q -k "reference_fixture" -p no:cacheprovider 2>&1 | tail -1
for nq in 2 4 2 4; do PMCE_GRU_NQ=$nq python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('NQ=$nq', d['value'],'clips/s', d['ms_per_step'],'ms  gru_step', d['kernel_ms_per_step']['gru_step'])"; done
