export PMCE_SYNTHETIC_BASE_DATA=1
for m in 1 2; do echo "== MINW=$m"; PMCE_SEQ_ATTN_MINW=$m python scripts/microbench/seqattn_ab.py 2>&1 | grep -v amdgpu; done
echo "== v1"; PMCE_SEQ_ATTN_V1=1 python scripts/microbench/seqattn_ab.py 2>&1 | grep -v amdgpu
