"""rocprofv3 --kernel-trace --stats summary (…kernel_stats.csv) -> per kernel CLASS totals, the classes bench.py reports.
usage: trace_classes.py <kernel_stats.csv> <steps the traced command ran (warm-up + timed + instrumented)> [bench json line]
With the bench line given, the roofline fraction is recomputed from the TRACE's average duration:
    frac = algorithmic_bytes_per_launch / avg_ns / 8 TB/s    (one division, both numbers in this file)."""
import csv
import json
import sys

import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from traffic_agg import classify


def main():
    path, steps = sys.argv[1], int(sys.argv[2])
    agg = {}
    for r in csv.DictReader(open(path)):
        c = agg.setdefault(classify(r["Name"]), {"calls": 0, "total_ns": 0})
        c["calls"] += int(r["Calls"])
        c["total_ns"] += int(r["TotalDurationNs"])
    out = {"source": f"rocprofv3 --kernel-trace --stats -- python bench.py ... ({path})", "steps_traced": steps, "classes": {}}
    for k, c in sorted(agg.items(), key=lambda kv: -kv[1]["total_ns"]):
        out["classes"][k] = {"calls": c["calls"], "calls_per_step": round(c["calls"] / steps, 1),
                             "total_ms": round(c["total_ns"] / 1e6, 3), "ms_per_step": round(c["total_ns"] / 1e6 / steps, 3),
                             "avg_us": round(c["total_ns"] / 1e3 / max(c["calls"], 1), 2)}
    if len(sys.argv) > 3:
        line = json.loads([l for l in open(sys.argv[3]).read().splitlines() if l.startswith("{")][-1])
        rl = line.get("roofline")
        if rl:
            k = rl["kernel"]
            avg_us = out["classes"][k]["avg_us"]
            if rl["bound"] == "hbm":
                ach = rl["algorithmic_bytes_per_launch"] / (avg_us * 1e-6) / 1e9
            else:
                ach = rl["algorithmic_flops_per_launch"] / (avg_us * 1e-6) / 1e12
            out["roofline_check"] = {
                "kernel": k, "bound": rl["bound"], "algorithmic_bytes_per_launch": rl["algorithmic_bytes_per_launch"],
                "algorithmic_flops_per_launch": rl["algorithmic_flops_per_launch"], "trace_avg_us": avg_us,
                "bench_avg_us": rl["avg_launch_us"], "achieved_from_trace": round(ach, 1), "unit": rl["unit"],
                "peak": rl["peak"], "frac_from_trace": round(ach / rl["peak"], 4), "frac_in_bench_line": rl["frac"]}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
