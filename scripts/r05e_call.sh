export TMPDIR=/tmp PMCE_SYNTHETIC_BASE_DATA=1
mkdir -p gpurun_out/r05e; O=gpurun_out/r05e
timeout 500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -m gpu -q -x -k "small_grid or gemm_modes or reference_fixture" -p no:cacheprovider > $O/pytest_small.log 2>&1; tail -3 $O/pytest_small.log | cut -c1-220
rm -f $O/ab_libs.txt
PMCE_ROUND=r05e bash scripts/ab_libs.sh "s0 new s2" "1 2 8 64" 256 512 2>&1 | cut -c1-60,200-330
