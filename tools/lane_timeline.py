#!/usr/bin/env python3
"""The PRODUCTION schedule's timeline from the event-pair dump of one instrumented step (bench.py with SFK_PER_LAYER=...: the
`.lanes` file; `t_ms` = start of every kernel on its own lane, `ms` = its duration there):
   python tools/lane_timeline.py gpurun_out/per_layer.json.lanes [gpurun_out/per_layer.json] [gap_us=150]
per lane: kernels, busy time, first start / last end, the idle gaps longer than gap_us with the kernels either side; with the
serial dump as second argument also the in-situ / alone-on-the-chip inflation per lane.  (Not a rocprofv3 trace: under the
tracer the host is the bottleneck and the lanes alternate -- DESIGN.md section 4d.)"""
import json
import sys
from collections import defaultdict

lanes = json.load(open(sys.argv[1]))
serial = json.load(open(sys.argv[2])) if len(sys.argv) > 2 and sys.argv[2].endswith(".json") else None
gap_us = float(sys.argv[-1]) if sys.argv[-1].replace(".", "").isdigit() else 150.0
end = max(r["t_ms"] + r["ms"] for r in lanes)
print(f"instrumented step: {end:.2f} ms from the first record to the end of the last kernel ({len(lanes)} kernels)")
ser = defaultdict(float)
if serial:
    for r in serial:
        ser[r["lane"]] += r["ms"]
for ln in sorted({r["lane"] for r in lanes}):
    rs = sorted((r for r in lanes if r["lane"] == ln), key=lambda r: r["t_ms"])
    busy = sum(r["ms"] for r in rs)
    gaps = [(b["t_ms"] - a["t_ms"] - a["ms"], a, b) for a, b in zip(rs, rs[1:])]
    big = [g for g in gaps if g[0] * 1e3 > gap_us]
    line = f"lane {ln}: {len(rs)} kernels, busy {busy:.2f} ms, first start {rs[0]['t_ms']:.2f}, last end {rs[-1]['t_ms'] + rs[-1]['ms']:.2f}"
    if serial:
        line += f"; alone on the chip {ser[ln]:.2f} ms -> x{busy / max(ser[ln], 1e-9):.2f} in situ"
    print(line + f"; {len(big)} gaps > {gap_us:.0f} us, {sum(g[0] for g in big):.2f} ms")
    for g, a, b in big[:12]:
        name = lambda r: (r["kind"] + " " + r.get("layer", "")[7:48]).strip()
        print(f"     {a['t_ms'] + a['ms']:6.2f} ms  idle {g * 1e3:5.0f} us   after {name(a)}   before {name(b)}")
