#!/usr/bin/env python
"""Per-kernel statistics of a rocprofv3 --kernel-trace CSV restricted to the WORKLOAD launches: bench.py first runs a tiny
set-up scene through the same kernels (code-object loading), which rocprofv3 --stats averages in.  For every kernel this
keeps the launches of its largest grid and reports count / average / min / max, next to the all-launch figures.
usage: tools/trace_workload_stats.py <kernel_trace.csv> <out.csv>"""
import csv
import re
import sys
from collections import defaultdict

rows = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    rows[r["Kernel_Name"]].append((grid, int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
out = []
for name, ls in rows.items():
    big = max(g for g, _ in ls)
    w = [d for g, d in ls if g == big]
    out.append((sum(w), name, len(ls), sum(d for _, d in ls) / len(ls), len(w), sum(w) / len(w), min(w), max(w)))
out.sort(reverse=True)
tot = sum(o[0] for o in out) or 1
with open(sys.argv[2], "w", newline="") as f:
    wr = csv.writer(f)
    wr.writerow(["Name", "AllCalls", "AllAverageNs", "WorkloadCalls", "WorkloadAverageNs", "WorkloadMinNs", "WorkloadMaxNs", "WorkloadPercentage"])
    for s, name, n_all, avg_all, n, avg, mn, mx in out:
        short = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", ""))[:90]
        wr.writerow([short, n_all, round(avg_all, 1), n, round(avg, 1), mn, mx, round(100.0 * s / tot, 2)])
print(f"{len(out)} kernels -> {sys.argv[2]}")
