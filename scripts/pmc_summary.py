"""FETCH_SIZE / WRITE_SIZE rocprofv3 passes (scripts/gpu_session.sh pmc) -> per-kernel HBM traffic per launch (KiB), the file
bench.py reads for `roofline.traffic`.  usage: pmc_summary.py <dir with FETCH_SIZE/ and WRITE_SIZE/> <out.json>"""
import collections
import csv
import glob
import json
import sys

root, out_path = sys.argv[1], sys.argv[2]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"{root}/{c}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != c:
                continue
            name = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:60]
            agg[name][0] += float(row["Counter_Value"])
            agg[name][1] += 1
    out[c] = {k: (v[0] / v[1], v[1]) for k, v in agg.items()}
res = {}
print("kernel | launches | FETCH_SIZE KiB/launch (raw) | WRITE_SIZE KiB/launch")
for n in sorted(set(out["FETCH_SIZE"]) | set(out["WRITE_SIZE"])):
    f = out["FETCH_SIZE"].get(n, (0, 0))
    w = out["WRITE_SIZE"].get(n, (0, 0))
    if "at::native" in n or "rocclr" in n:
        continue
    print(f"{n:60s} {f[1]:5d} {f[0]:14.1f} {w[0]:14.1f}")
    res[n] = {"launches": f[1], "fetch_kib_raw": f[0], "write_kib": w[0]}
# which build these counters belong to (bench.py flags the file as stale when the loaded library's id differs)
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from pmce_amd import build as _b
res["_meta"] = {"build_id": _b.source_id(), "passes": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate runs, kernel-trace only), KiB per launch"}
json.dump(res, open(out_path, "w"), indent=1)
