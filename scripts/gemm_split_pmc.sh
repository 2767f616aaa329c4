#!/bin/bash
# PMC counters of the split-f16 GEMM at two lifter shapes (one rocprofv3 pass per counter group; kernel-trace only)
set -u
export TMPDIR=/tmp
rm -rf gpurun_out/spmc; mkdir -p gpurun_out/spmc
cat > /tmp/one_split.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["REPO"])
import torch
from pmce_amd import ops
dev = torch.device("cuda:0")
for (M, N, K, act, res) in [(69632, 1536, 512, 0, False), (69632, 512, 512, 0, True)]:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev)
    R = torch.randn(M, N, device=dev) if res else None
    Wp, ws = ops.pack_split_f16(W)
    out = torch.empty(M, N, device=dev)
    for _ in range(6): ops.gemm_nt_split(A, Wp, ws, b, R, act, out=out)
    torch.cuda.synchronize()
PY
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" "SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum" "TCC_REQ_sum TCC_READ_sum TCC_EA_RDREQ_32B_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_INSTS_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH"; do
  i=$((i+1))
  (cd /tmp && REPO=$OLDPWD timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OLDPWD/gpurun_out/spmc/g$i -o pmc -- python /tmp/one_split.py > $OLDPWD/gpurun_out/spmc/g$i.log 2>&1)
  echo "group $i exit $?"
done
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("gpurun_out/spmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        if "gemm_nt" not in name and "gemm_split" not in name: continue
        key = name.split("(")[0].replace("void gemm_nt_kernel", "gemm").replace("void gemm_split_kernel", "split") + f" grid={row.get('Grid_Size','?')}"
        a = agg[key][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
for k in sorted(agg):
    print(k)
    for c, (v, n) in sorted(agg[k].items()):
        print(f"    {c:32s} {v/n:16.1f}  (n={n})")
PY
find gpurun_out/spmc -name "*.csv" -size +4M -delete
