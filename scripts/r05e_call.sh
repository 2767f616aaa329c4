export TMPDIR=/tmp PMCE_SYNTHETIC_BASE_DATA=1
mkdir -p gpurun_out/r05e; O=gpurun_out/r05e
rm -f $O/ab_libs.txt
PMCE_ROUND=r05e bash scripts/ab_libs.sh "fa0 new" "1 64 256" 256 512 2>&1 | cut -c1-30,250-420
