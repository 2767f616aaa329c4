export PMCE_SYNTHETIC_BASE_DATA=1
Q="--no-variant --no-cpu-baseline --no-latency --no-host-fed --steps 20 --warmup 5 --windows 3 --sustained-seconds 0"
run() { tag=$1; shift; timeout 200 python bench.py $Q --detail-file /tmp/d.json "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$tag', d['value'], d['ms_per_step'], d['config'].get('lanes'))"; }
for rep in 1 2; do
for d in 1 2 3 4; do
run c512_b64_depth$d --batch 64 --pipeline-depth $d
run c512_b16_depth$d --batch 16 --pipeline-depth $d
run c256_j19_b128_depth$d --embed-dim 256 --joints 19 --batch 128 --pipeline-depth $d
done
run c256_b256_depth3 --embed-dim 256 --pipeline-depth 3
run c256_b256_depth2 --embed-dim 256 --pipeline-depth 2
run c512_b256_depth2
run c512_b256_depth3 --pipeline-depth 3
done
