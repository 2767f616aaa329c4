export PMCE_ROUND=r05 PMCE_SYNTHETIC_BASE_DATA=1
mkdir -p gpurun_out/r05
python -c "import pmce_amd.build as b; print(b.build())" > gpurun_out/r05/build.log 2>&1
rm -f gpurun_out/r05/ab_libs.txt
bash scripts/ab_libs.sh "new u2" "256" 256
