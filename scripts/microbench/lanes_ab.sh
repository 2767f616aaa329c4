Q="--no-variant --no-cpu-baseline --no-latency --no-host-fed --steps 20 --warmup 5 --windows 3"
for i in 1 2; do
for C in 512 256; do
for f in "" "--no-stagger" "--pipeline-depth 3" "--pipeline-depth 1"; do
  timeout 300 python bench.py $Q --embed-dim $C $f 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('C=$C flags=[$f]', d['value'], d['ms_per_step'])"
done; done; done
