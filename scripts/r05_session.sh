export PMCE_ROUND=r05 PMCE_SYNTHETIC_BASE_DATA=1
mkdir -p gpurun_out/r05
python -c "import pmce_amd.build as b; print(b.build())" > gpurun_out/r05/build.log 2>&1
rm -f gpurun_out/r05/ab_libs.txt
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "adaln_mlp or cross_attn or coevo or vertex_self_attn or never_returns or nonfinite or fixture or joint or decoder or full_size" > gpurun_out/r05/pytest_quick_new.log 2>&1
echo "quick tests exit $?"; tail -n 5 gpurun_out/r05/pytest_quick_new.log; grep -n "fused CrossAttentionBlock\|scaled weights" gpurun_out/r05/pytest_quick_new.log | head -12
bash scripts/ab_libs.sh "new" "1 256" 256
