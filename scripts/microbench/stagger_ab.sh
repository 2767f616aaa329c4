export PMCE_SYNTHETIC_BASE_DATA=1
Q="--no-variant --no-cpu-baseline --no-latency --no-host-fed --steps 20 --warmup 5 --windows 3 --sustained-seconds 0"
run() { tag=$1; shift; timeout 200 python bench.py $Q --detail-file /tmp/d.json "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$tag', d['value'], d['ms_per_step'])"; }
for rep in 1 2 3; do
run c512_stagger
run c512_free --no-stagger
run c256_stagger --embed-dim 256
run c256_free --embed-dim 256 --no-stagger
run c512_b64_stagger --batch 64
run c512_b64_free --batch 64 --no-stagger
run c256_j19_b128_stagger --embed-dim 256 --joints 19 --batch 128
run c256_j19_b128_free --embed-dim 256 --joints 19 --batch 128 --no-stagger
done
