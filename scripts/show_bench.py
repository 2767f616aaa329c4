import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l)
        print(d["value"], "clips/s", d["ms_per_step"], "ms/step", d["roofline"])
        print({k: v for k, v in d["kernel_ms_per_step"].items() if v > 0.04})
        print("cpu:", d.get("cpu_baseline"))
    else:
        print(l.strip()[:200])
