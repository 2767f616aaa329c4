export PMCE_ROUND=r05 PMCE_SYNTHETIC_BASE_DATA=1
mkdir -p gpurun_out/r05
python -c "import pmce_amd.build as b; print(b.build())" > gpurun_out/r05/build.log 2>&1
rm -f gpurun_out/r05/ab_libs.txt
timeout 900 python -m pytest tests -m gpu -q -x -s -p no:cacheprovider -k "joint_stream or coevo or fixture or full_size or scripts or row_scaled or blocked" > gpurun_out/r05/pytest_quick_new.log 2>&1
echo "quick tests exit $?"; tail -n 4 gpurun_out/r05/pytest_quick_new.log; grep "tokens_kv vs" gpurun_out/r05/pytest_quick_new.log
bash scripts/ab_libs.sh "new" "1 256" 256
