for g in 1 2 3 4; do echo "== persistent workgroups per CU: $g"; PMCE_GEMM_GRID=$g python scripts/gemm_sweep.py 2>&1 | grep -v amdgpu.ids | head -5; done
echo "== default"; python scripts/gemm_sweep.py 2>&1 | grep -v amdgpu.ids
