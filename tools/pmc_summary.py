"""Sum rocprofv3 --pmc counter_collection CSVs per (kernel prefix, counter): python tools/pmc_summary.py <dir> [prefix ...]"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
pref = sys.argv[2:] or ["blend_bwd", "blend_fwd"]
acc = defaultdict(lambda: [0.0, 0])
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for p in pref:
            if p in k:
                a = acc[(p, r["Counter_Name"])]
                a[0] += float(r["Counter_Value"]); a[1] += 1
for (p, c), (v, n) in sorted(acc.items()):
    print(f"{p:14s} {c:28s} per-launch {v / n:16.1f}   (n={n})")
