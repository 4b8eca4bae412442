#!/usr/bin/env python
"""Instruction mix per basic block of one kernel in a `hipcc -S` listing:
   python tools/isa_blocks.py blend.s <mangled-name substring> [min instructions] [dump label]"""
import collections
import re
import sys

src = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
minins = int(sys.argv[3]) if len(sys.argv) > 3 else 60
dump = sys.argv[4] if len(sys.argv) > 4 else None
start = next(i for i, l in enumerate(src) if re.match(r"^_Z\w*" + re.escape(key) + r"\w*:", l))
end = next(i for i in range(start, len(src)) if src[i].startswith(".Lfunc_end"))
blocks, cur = [], ["entry", []]
blocks.append(cur)
for l in src[start + 1:end]:
    t = l.strip()
    if re.match(r"^\.LBB\d+_\d+:", t):
        cur = [t.split(":")[0], []]
        blocks.append(cur)
    elif t and not t.startswith(";") and not t.startswith("."):
        cur[1].append(t)


def cat(ins):
    op = ins.split()[0]
    if op.startswith("v_mfma"): return "mfma"
    if "dpp" in ins or op.startswith("v_permlane") or op.startswith("v_readlane") or op.startswith("v_readfirst"): return "valu_x"
    if op in ("v_exp_f32", "v_rcp_f32", "v_log_f32", "v_sqrt_f32", "v_rsq_f32"): return "trans"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_")): return "vmem"
    if op.startswith("scratch_"): return "scratch"
    return "other"


tot = collections.Counter()
for name, ins in blocks:
    c = collections.Counter(cat(i) for i in ins)
    tot.update(c)
    if len(ins) >= minins:
        print(f"{name:12s} {len(ins):5d}", dict(sorted(c.items())))
    if dump and name == dump:
        print("\n".join(ins))
print("total", sum(tot.values()), dict(sorted(tot.items())))
