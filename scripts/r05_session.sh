export PMCE_ROUND=r05
bash scripts/gpu_session.sh tests bench ab_r4 prof pmc counters
