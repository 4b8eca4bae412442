#!/usr/bin/env python
"""Timeline of a rocprofv3 --kernel-trace CSV: for the last `n` launches, kernel name, duration and the idle gap before it
(start - previous end); then busy / idle totals.  usage: tools/gap_stats.py <kernel_trace.csv> [n]"""
import csv
import re
import sys
from collections import defaultdict

rs = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
rs = rs[-n:]
busy = idle = 0
per = defaultdict(lambda: [0, 0, 0])
prev = None
for r in rs:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", ""))
    name = re.sub(r"<.*", "", name)[:40]
    gap = s - prev if prev is not None else 0
    prev = max(e, prev or 0)
    busy += e - s
    idle += max(gap, 0)
    p = per[name]
    p[0] += 1; p[1] += e - s; p[2] += max(gap, 0)
print(f"launches {len(rs)} busy {busy / 1e3:.1f} us idle {idle / 1e3:.1f} us")
for name, (c, d, g) in sorted(per.items(), key=lambda x: -x[1][1]):
    print(f"{name:42s} n {c:5d} avg {d / c / 1e3:8.2f} us  gap before {g / c / 1e3:7.2f} us")
