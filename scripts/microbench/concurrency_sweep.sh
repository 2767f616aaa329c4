export PMCE_SYNTHETIC_BASE_DATA=1
Q="--no-variant --no-cpu-baseline --no-latency --no-host-fed --steps 20 --warmup 5 --windows 3 --sustained-seconds 0"
run() { tag=$1; shift; timeout 200 python bench.py $Q --detail-file /tmp/d.json "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$tag', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
run depth2 --pipeline-depth 2
run depth1 --pipeline-depth 1
run depth3 --pipeline-depth 3
run depth4 --pipeline-depth 4
run depth2_nostagger --pipeline-depth 2 --no-stagger
run depth3_nostagger --pipeline-depth 3 --no-stagger
run single_stream --single-stream
GPU_MAX_HW_QUEUES=8 run depth2_hwq8 --pipeline-depth 2
GPU_MAX_HW_QUEUES=8 run depth3_hwq8 --pipeline-depth 3
run c256_depth2 --embed-dim 256 --pipeline-depth 2
run c256_depth3 --embed-dim 256 --pipeline-depth 3
run c256_depth2_nostagger --embed-dim 256 --pipeline-depth 2 --no-stagger
done
