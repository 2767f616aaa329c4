#!/bin/bash
# HBM traffic per kernel from PMC counters (separate passes, kernel-trace only - see MI355X_MICROARCH.md §HBM):
#   FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-counts wide coalesced reads by 2x.
set -u
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc/$c
  (cd /tmp && timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OLDPWD/gpurun_out/pmc/$c -o pmc -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --single-stream > $OLDPWD/gpurun_out/pmc/$c.log 2>&1)
  echo "$c exit $?"
done
python - <<'PY'
import csv, glob, collections, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(f"gpurun_out/pmc/{c}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != c: continue
            name = row["Kernel_Name"].split("(")[0][:60]
            agg[name][0] += float(row["Counter_Value"]); agg[name][1] += 1
    out[c] = {k: (v[0] / v[1], v[1]) for k, v in agg.items()}
names = sorted(set(out["FETCH_SIZE"]) | set(out["WRITE_SIZE"]))
print("kernel | launches | FETCH_SIZE KiB/launch (raw) | WRITE_SIZE KiB/launch")
res = {}
for n in names:
    f = out["FETCH_SIZE"].get(n, (0, 0)); w = out["WRITE_SIZE"].get(n, (0, 0))
    print(f"{n:60s} {f[1]:5d} {f[0]:14.1f} {w[0]:14.1f}")
    res[n] = {"launches": f[1], "fetch_kib_raw": f[0], "write_kib": w[0]}
json.dump(res, open("gpurun_out/pmc/summary.json", "w"), indent=1)
PY
find gpurun_out/pmc -name "*.csv" -size +8M -delete
