"""rocprofv3 --pmc counter_collection CSVs (several passes) of tools/blend_bench.py -> one JSON with per-launch counters of
the two compositing kernels and the utilisation figures derived from them.
   python tools/pmc_blend_counters.py <out.json> <pass dir> [<pass dir> ...]"""
import csv
import glob
import json
import sys
from collections import defaultdict

import os
import re
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from stamp import stamp  # noqa: E402

out, dirs = sys.argv[1], sys.argv[2:]
FULL = {}   # kernel -> full template name of its largest launches
KERNELS = {"blend_fwd_kernel": "blend_fwd", "blend_bwd_mfma_kernel": "blend_bwd", "blend_bwd_quarter_kernel": "blend_bwd", "blend_bwd_wide_quarter_kernel": "blend_bwd", "blend_bwd_sets_kernel": "blend_bwd_sets", "blend_bwd_sets_quarter_kernel": "blend_bwd_sets"}
rows = []
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            for k in KERNELS:
                if k in r["Kernel_Name"]:
                    g = int(r.get("Grid_Size", 0) or 0)
                    rows.append((k, g, r["Counter_Name"], float(r["Counter_Value"])))
                    if g >= FULL.get(k, (0, ""))[0]:
                        FULL[k] = (g, re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")))
big = defaultdict(int)          # only the largest launches of a kernel are the workload (a tiny set-up scene runs first)
for k, g, _, _ in rows:
    big[k] = max(big[k], g)
acc = defaultdict(lambda: [0.0, 0])
for k, g, c, v in rows:
    if g == big[k]:
        a = acc[(k, c)]
        a[0] += v; a[1] += 1
import os
res = dict(stamp())
res.update({"source": os.environ.get("PMC_SOURCE", "rocprofv3 --pmc, three passes, per launch (largest launches of each kernel), summed over the 8 "
                 "XCDs; SQ_ACTIVE_INST_* / SQ_WAIT_* / SQ_WAVE_CYCLES in quad-cycles (guides/MI355X_MICROARCH.md)"),
       "config": os.environ.get("PMC_CONFIG"),
       "kernels": {}})
SIMDS, WAVE_SLOTS, XCDS = 1024, 256 * 16, 8
for k in KERNELS:
    c = {n: round(v / cnt, 1) for (kk, n), (v, cnt) in sorted(acc.items()) if kk == k}
    if not c:
        continue
    cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / XCDS          # kernel duration in shader clocks
    d = {}
    if cyc:
        if "SQ_ACTIVE_INST_VALU" in c:
            d["valu_busy_frac"] = round(4.0 * c["SQ_ACTIVE_INST_VALU"] / (SIMDS * cyc), 3)   # SIMD-cycles issuing VALU / all SIMD-cycles
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            d["mfma_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (SIMDS * cyc), 3)
        if "SQ_WAVE_CYCLES" in c:
            # average resident waves per SIMD over the kernel's life (the kernel's own limit: 4 for blend_bwd by registers / LDS,
            # 7 for blend_fwd); wave_slot_residency = the same relative to 4 waves per SIMD (16 per CU), kept for comparison
            # with the round-1 record
            d["waves_per_simd"] = round(4.0 * c["SQ_WAVE_CYCLES"] / (SIMDS * cyc), 3)
            d["wave_slot_residency"] = round(4.0 * c["SQ_WAVE_CYCLES"] / (WAVE_SLOTS * cyc), 3)
        d["duration_cycles"] = round(cyc, 1)
    res["kernels"][KERNELS[k]] = {"kernel_name": FULL.get(k, (0, None))[1], "counters": c, "derived": d}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: v["derived"] for k, v in res["kernels"].items()}))
