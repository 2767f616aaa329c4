mkdir -p gpurun_out/r03
for ns in 3 4 5; do
  PMCE_EXTRA_HIPCC_FLAGS="-DPMCE_WS_ABLATE -DWS_NS_OVERRIDE=$ns" python -m pmce_amd.build --force > gpurun_out/r03/build_ns.log 2>&1
  echo "== ring of $ns stages" >> gpurun_out/r03/gemm_ws_ns_sweep.txt
  timeout 200 python scripts/microbench/gemm_ws.py --check-opt 2>&1 | grep "OPT=0\|OPT=1" >> gpurun_out/r03/gemm_ws_ns_sweep.txt
done
cat gpurun_out/r03/gemm_ws_ns_sweep.txt
