"""Per-kernel HBM/fabric traffic per launch from two rocprofv3 --pmc passes (GPU box):
   read : TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum  -> 32 n32 + 64 n64 + 128 n128
          (n64 = RDREQ - n32 - n128 when the 64B counter is unavailable)
   write: WRITE_SIZE (KiB)
usage: python tools/pmc_traffic.py <read_pass_dir> <write_pass_dir> <out.json> "<source note>" [<bench config tag>]
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from stamp import stamp  # noqa: E402

FULL = {}   # short kernel name -> the full (template) name of its largest launches


def load(d):
    """per kernel: counters averaged over its LARGEST launches only (the bench also runs a tiny set-up scene through the
    same kernels: those launches are not the workload)"""
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
            if m:
                rows.append((m.group(1), int(r.get("Grid_Size", 0) or 0), r["Counter_Name"], float(r["Counter_Value"])))
                g = int(r.get("Grid_Size", 0) or 0)
                if g >= FULL.get(m.group(1), (0, ""))[0]:
                    FULL[m.group(1)] = (g, re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")))
    big = defaultdict(int)
    for k, g, _, _ in rows:
        big[k] = max(big[k], g)
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for k, g, c, v in rows:
        if g == big[k]:
            a = acc[k][c]
            a[0] += v; a[1] += 1
    return {k: {c: v[0] / v[1] for c, v in cs.items()} for k, cs in acc.items()}


rd, wr = load(sys.argv[1]), load(sys.argv[2])
out = dict(stamp(), source=sys.argv[4], config=sys.argv[5] if len(sys.argv) > 5 else None, kernels={})
for k in sorted(set(rd) | set(wr)):
    c = rd.get(k, {})
    n32, n64, n128 = c.get("TCC_EA0_RDREQ_32B_sum", 0.0), c.get("TCC_EA0_RDREQ_64B_sum", 0.0), c.get("TCC_EA0_RDREQ_128B_sum", 0.0)
    rb = 32 * n32 + 64 * n64 + 128 * n128
    wb = wr.get(k, {}).get("WRITE_SIZE", 0.0) * 1024
    out["kernels"][k] = {"read_bytes": int(rb), "write_bytes": int(wb), "total_MB": round((rb + wb) / 1e6, 1),
                         "kernel_name": FULL.get(k, (0, None))[1]}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
