#!/usr/bin/env python3
"""Dispatches of one kernel + grid that take much longer than their siblings, from a rocprofv3 kernel trace csv:
   python tools/trace_outliers.py <kernel_trace.csv> [ratio=2.0]
The same (kernel, grid, workgroup) launched many times per step should take about the same time each time; a dispatch
RATIO x over its group's median is a cold-start / contention artefact or a layer that deserves a look."""
import csv
import statistics
import sys
from collections import defaultdict

path = sys.argv[1]
ratio = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
groups = defaultdict(list)
with open(path) as f:
    for r in csv.DictReader(f):
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        key = (r["Kernel_Name"][:110], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")))
        groups[key].append((dur, int(r["Start_Timestamp"])))
rows = []
for key, xs in groups.items():
    if len(xs) < 4:
        continue
    med = statistics.median(d for d, _ in xs)
    slow = [d for d, _ in xs if d > ratio * med and d - med > 20.0]
    if slow:
        rows.append((sum(slow) - med * len(slow), key, len(xs), med, len(slow), max(slow)))
rows.sort(reverse=True)
print(f"{'excess us':>10s} {'n':>6s} {'median':>8s} {'slow':>5s} {'max':>8s}  kernel [grid/wg]")
for ex, key, n, med, ns, mx in rows[:40]:
    print(f"{ex:10.1f} {n:6d} {med:8.1f} {ns:5d} {mx:8.1f}  {key[0]} [{key[1]}/{key[2]}]")
