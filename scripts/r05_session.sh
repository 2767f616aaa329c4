export PMCE_ROUND=r05 PMCE_SYNTHETIC_BASE_DATA=1
bash scripts/gpu_session.sh tests
bash scripts/ab_libs.sh "new" "1 256" 256 512
