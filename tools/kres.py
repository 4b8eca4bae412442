#!/usr/bin/env python
"""Kernel resource table from `hipcc -Rpass-analysis=kernel-resource-usage` remarks (stderr of a -S / -c compile):
   python tools/kres.py remarks.txt [name filter ...]   ->  name | VGPR | AGPR | scratch B/lane | occupancy | LDS bytes"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
filt = sys.argv[2:]
blocks = re.split(r"remark: Function Name: ", txt)[1:]
names = [b.split(" [")[0].strip() for b in blocks]
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
for b, d in zip(blocks, dem):
    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1
    d = re.sub(r"\(BlendArgs\)|\(anonymous namespace\)::", "", d)
    if filt and not any(f in d for f in filt):
        continue
    v, ag, sc, oc = g("    VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")
    lds, sg = g(r"LDS Size \[bytes/block\]"), g("TotalSGPRs")
    print(f"{d[:70]:70s} vgpr {v:3d} agpr {ag:3d} scratch {sc:4d} occ {oc} lds {lds:6d} sgpr {sg:3d}")
