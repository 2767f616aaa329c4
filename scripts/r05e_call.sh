export TMPDIR=/tmp PMCE_SYNTHETIC_BASE_DATA=1
mkdir -p gpurun_out/r05e; O=gpurun_out/r05e
timeout 500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -m gpu -q -x -k "small_batch or gemm_modes or reference_fixture or gru_all" -p no:cacheprovider > $O/pytest_small.log 2>&1; tail -3 $O/pytest_small.log | cut -c1-220
rm -f $O/ab_kernels3.txt
for i in 1 2; do AB_CLASSES=gru_step,gemm_gru_in,gemm_lifter AB_C=256 timeout 300 python scripts/microbench/ab_kernels.py 1 64 256 >> $O/ab_kernels3.txt 2>&1; done
cat $O/ab_kernels3.txt
