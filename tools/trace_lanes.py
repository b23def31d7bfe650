#!/usr/bin/env python3
"""Per hardware queue of one traced training step: kernels, busy time, span, idle gaps and what sits either side of the long
ones.   python tools/trace_lanes.py <kernel_trace.csv> [step_index=5] [gap_us=20]
A step = from the end of one step's last adam launch to the end of the next one's."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 5
big = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
short = lambda s: s.replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")[:52]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Queue_Id"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Grid_Size_X"], int(r["Dispatch_Id"])))
rows.sort(key=lambda r: r[1])
adam = [r for r in rows if "adam_kernel" in r[3]]
per_step = 2 if len(adam) % 2 == 0 else 1
ends = [adam[i][2] for i in range(per_step - 1, len(adam), per_step)]
t0, t1 = ends[which], ends[which + 1]
print(f"step {which}: {(t1 - t0) / 1e6:.3f} ms")
win = [r for r in rows if r[1] >= t0 and r[2] <= t1]
byq = defaultdict(list)
for r in win:
    byq[r[0]].append(r)
for q, rs in sorted(byq.items()):
    busy = sum(r[2] - r[1] for r in rs) / 1e6
    gaps = [((b[1] - a[2]) / 1e3, a, b) for a, b in zip(rs, rs[1:])]
    small = sum(g for g, _, _ in gaps if 0 < g < big) / 1e3
    large = [(g, a, b) for g, a, b in gaps if g >= big]
    print(f"queue {q}: {len(rs)} kernels, busy {busy:.2f} ms, first start {(rs[0][1] - t0) / 1e6:.2f}, last end {(rs[-1][2] - t0) / 1e6:.2f}; "
          f"gaps < {big:.0f} us sum {small:.2f} ms; {len(large)} longer gaps sum {sum(g for g, _, _ in large) / 1e3:.2f} ms")
    for g, a, b in large:
        print(f"    {(a[2] - t0) / 1e6:7.2f} ms  idle {g:7.0f} us   after {a[3]} [{a[4]}]   before {b[3]} [{b[4]}]")
