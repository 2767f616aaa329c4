#!/bin/bash
# PMC counters of vertex_sa (fp32 and split-f16 form) at B = 256: where do its cycles go (one rocprofv3 pass per counter group)
set -u
export TMPDIR=/tmp
rm -rf gpurun_out/vsapmc; mkdir -p gpurun_out/vsapmc
cat > /tmp/one_vsa.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["REPO"])
import torch
from pmce_amd import _lib
P = _lib.ptr
lib = _lib.load()
dev = torch.device("cuda:0")
B = 256
x = torch.randn(B, 431, 64, device=dev); qkv = torch.randn(B, 431, 192, device=dev) * 1.5
W = torch.randn(64, 64, device=dev) / 8; b = torch.randn(64, device=dev); y = torch.empty_like(x)
for f16 in (0, 1):
    for _ in range(5):
        _lib.check(lib.pmce_vertex_sa_ex_f32(P(x), P(qkv), P(W), P(b), P(y), B, f16, None), "vsa")
    torch.cuda.synchronize()
PY
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY"; do
  i=$((i+1))
  (cd /tmp && REPO=$OLDPWD timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OLDPWD/gpurun_out/vsapmc/g$i -o pmc -- python /tmp/one_vsa.py > $OLDPWD/gpurun_out/vsapmc/g$i.log 2>&1)
  echo "group $i exit $?"
done
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/vsapmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        if "vertex_sa" not in name: continue
        key = name.split("(")[0]
        a = agg[key][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
for f in glob.glob("gpurun_out/vsapmc/g1/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "vertex_sa" in row["Kernel_Name"]:
            dur[row["Kernel_Name"].split("(")[0]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
for k in sorted(agg):
    print(k, "us per launch (under the counters):", [round(d, 1) for d in dur.get(k, [])])
    for c, (v, n) in sorted(agg[k].items()):
        print(f"    {c:32s} {v/n:16.1f}  (n={n})")
PY
find gpurun_out/vsapmc -name "*.csv" -size +4M -delete
