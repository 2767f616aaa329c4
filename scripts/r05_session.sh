export PMCE_ROUND=r05
bash scripts/gpu_session.sh tests
bash scripts/ab_libs.sh "r4 gelu new" "1 256" 256 512
