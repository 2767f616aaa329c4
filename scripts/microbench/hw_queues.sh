export PMCE_SYNTHETIC_BASE_DATA=1
for q in 2 4 8 16; do for C in 512 256; do
echo -n "GPU_MAX_HW_QUEUES=$q C=$C: "; GPU_MAX_HW_QUEUES=$q python bench.py --embed-dim $C --steps 10 --windows 3 --no-cpu-baseline --no-host-fed --no-latency --no-variant 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['windows']['ms_per_step'])"
done; done
echo "default:"; for C in 512 256; do python bench.py --embed-dim $C --steps 10 --windows 3 --no-cpu-baseline --no-host-fed --no-latency --no-variant 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print(d['config']['embed_dim'], d['value'], d['windows']['ms_per_step'])"; done
