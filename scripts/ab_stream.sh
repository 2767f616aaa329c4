bash scripts/gpu_check.sh tests 2>&1 | tail -3
for i in 1 2; do
echo "== two streams"; python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python scripts/show_bench.py | head -1
echo "== single stream"; PMCE_SINGLE_STREAM=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python scripts/show_bench.py | head -1
done
