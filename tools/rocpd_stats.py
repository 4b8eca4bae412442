#!/usr/bin/env python
"""Dump the per-kernel summary of a rocprofv3 (rocpd sqlite) result DB as CSV.
usage: tools/rocpd_stats.py <results.db> <out.csv>"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
    for r in rows:
        w.writerow([r[0], r[1], round(r[2], 3), round(r[3], 3), round(r[4], 3)])
print(f"{len(rows)} kernels -> {sys.argv[2]}")
