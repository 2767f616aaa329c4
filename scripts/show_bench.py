"""Pretty-print bench.py JSON lines (from a file argument, or stdin when no argument is given)."""
import json
import sys

src = open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin


def rec(tag, d):
    print(f"[{tag}] {d['value']} clips/s  {d['ms_per_step']} ms/step  windows {d.get('windows', {}).get('ms_per_step')}")
    print("   roofline:", {k: v for k, v in (d.get("roofline") or {}).items() if k not in ("classes", "traffic_source")})
    print("   north-star:", d.get("roofline_cross_attention"))
    print("   attention :", d.get("roofline_attention"))
    print("   kernels (ms/step):", {k: v for k, v in (d.get("kernel_ms_per_step") or {}).items() if v > 0.04},
          "sum", d.get("kernel_ms_total_single_stream"))
    print("   cpu:", d.get("cpu_baseline"))


for l in src:
    if l.startswith("{"):
        d = json.loads(l)
        rec(f"C={d['config']['embed_dim']} n_gpus={d['n_gpus']}", d)
        print("   host_fed:", d.get("host_fed"))
        print("   latency:", d.get("latency"))
        for k, v in d.items():
            if k.startswith("variant_") and v:
                rec(k, v)
            if k.startswith("config_") and v:
                print(f"[{k}]", {kk: vv for kk, vv in v.items() if kk not in ("roofline", "cross_attention", "kernel_ms_per_step", "kernel_ms_per_window_batch", "what", "measured_by")})
                print("   roofline:", {kk: vv for kk, vv in (v.get("roofline") or {}).items() if kk in ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_launch_ms", "launches_per_step")})
    else:
        print(l.strip()[:200])
