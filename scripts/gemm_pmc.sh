#!/bin/bash
# PMC counters of the lifter GEMM shapes (one rocprofv3 pass per counter group; kernel-trace only)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/gpmc
cat > /tmp/one_gemm.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["REPO"])
import torch
from pmce_amd import ops
dev = torch.device("cuda:0")
for (M, N, K, act, res) in [(69632, 768, 256, 0, False), (69632, 256, 256, 0, True), (69632, 512, 256, 1, False), (4096, 6144, 2048, 0, False),
                            (69632, 1536, 512, 0, False), (69632, 512, 512, 0, True), (69632, 1024, 512, 1, False), (69632, 512, 1024, 0, True)]:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev)
    R = torch.randn(M, N, device=dev) if res else None
    out = torch.empty(M, N, device=dev)
    for _ in range(6): ops.gemm_nt(A, W, b, R, act, out=out)
    torch.cuda.synchronize()
PY
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  (cd /tmp && REPO=$OLDPWD timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OLDPWD/gpurun_out/gpmc/g$i -o pmc -- python /tmp/one_gemm.py > $OLDPWD/gpurun_out/gpmc/g$i.log 2>&1)
  echo "group $i exit $?"
done
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("gpurun_out/gpmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        if "gemm_nt" not in name: continue
        key = name.split("(")[0].replace("void gemm_nt_kernel", "gemm") + f" grid={row.get('Grid_Size','?')} lds={row.get('LDS_Block_Size','?')}"
        a = agg[key][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
for k in sorted(agg):
    print(k)
    for c, (v, n) in sorted(agg[k].items()):
        print(f"    {c:32s} {v/n:16.1f}  (n={n})")
PY
find gpurun_out/gpmc -name "*.csv" -size +4M -delete
