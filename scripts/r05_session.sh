export PMCE_ROUND=${PMCE_ROUND:-r05f}
bash scripts/gpu_session.sh tests prof pmc counters install bench
