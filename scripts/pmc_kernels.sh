#!/bin/bash
# PMC counters for selected kernels of the forward (one pass per counter group); usage: [PMCE_PMC_C=256] pmc_kernels.sh "regex"
# summary (with the build id of the library the counters belong to): gpurun_out/kpmc/summary.json
set -u
pat=${1:-vertex_ca}
export TMPDIR=/tmp
rm -rf gpurun_out/kpmc
mkdir -p gpurun_out/kpmc
i=0
# PMCE_PMC_ONLY=3 runs only the third group (LDS counters)
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM"; do
  i=$((i+1))
  if [[ -n "${PMCE_PMC_ONLY:-}" && "$i" != "$PMCE_PMC_ONLY" ]]; then continue; fi
  (cd /tmp && timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OLDPWD/gpurun_out/kpmc/g$i -o pmc -- python $OLDPWD/bench.py --embed-dim ${PMCE_PMC_C:-512} --steps 1 --warmup 1 --windows 1 --no-cpu-baseline --no-latency --no-host-fed --no-variant --single-stream > $OLDPWD/gpurun_out/kpmc/g$i.log 2>&1)
done
PAT="$pat" python - <<'PY'
import csv, glob, collections, os, re
pat = re.compile(os.environ["PAT"])
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("gpurun_out/kpmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        if not pat.search(name): continue
        a = agg[name[:70]][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
import json
out = {}
for k in sorted(agg):
    print(k)
    for c, (v, n) in sorted(agg[k].items()):
        print(f"    {c:32s} {v/n:16.1f}  (n={n})")
    d = {c: v / n for c, (v, n) in agg[k].items()}
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs
    if d.get("GRBM_GUI_ACTIVE") and d.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        d["mfma_busy_fraction"] = round((d["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024) / (d["GRBM_GUI_ACTIVE"] / 8), 4)
        print(f"    -> matrix pipe busy {100 * d['mfma_busy_fraction']:.1f} % of the kernel's cycles")
    out[k] = d
import sys
sys.path.insert(0, os.getcwd())
from pmce_amd import build as _b
out["_meta"] = {"build_id": _b.source_id(), "embed_dim": int(os.environ.get("PMCE_PMC_C", "512")),
                "command": "bench.py --embed-dim C --steps 1 --warmup 1 --windows 1 --single-stream under rocprofv3 --pmc <group> --kernel-trace (one pass per group)"}
json.dump(out, open("gpurun_out/kpmc/summary.json", "w"), indent=1)
PY
find gpurun_out/kpmc -name "*.csv" -size +4M -delete
