export PMCE_SYNTHETIC_BASE_DATA=1
for rep in 1 2 3; do
for s in "" "--stagger"; do
  timeout 200 python scripts/eval_sharded.py --clips 35515 --joints 19 --min-seconds 3 $s 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('eval_j19 [$s]', d.get('clips_per_s_incl_metrics'), d.get('pass_seconds_min_max'))"
  timeout 300 python bench.py --no-variant --no-cpu-baseline --no-latency --steps 20 --warmup 5 --windows 3 --sustained-seconds 0 --detail-file /tmp/d.json $s 2>/dev/null | tail -1 > /dev/null; python -c "
import json; d=json.load(open('/tmp/d.json')); print('bench [$s] value', d['value'], 'host_fed', d['host_fed']['value'])"
done; done
