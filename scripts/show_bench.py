"""Pretty-print bench.py JSON lines (from a file argument, or stdin when no argument is given)."""
import json
import sys

src = open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin
for l in src:
    if l.startswith("{"):
        d = json.loads(l)
        print(d["value"], "clips/s", d["ms_per_step"], "ms/step", d["roofline"])
        print({k: v for k, v in d["kernel_ms_per_step"].items() if v > 0.04})
        print("cpu:", d.get("cpu_baseline"))
    else:
        print(l.strip()[:200])
