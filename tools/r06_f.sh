#!/bin/bash
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r06f
mkdir -p $O
export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --per-frame --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-extra-lines --no-other-configs"
(cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o bench -- $B > $GRAFT_REPO_ROOT/$O/prof_stdout.log 2>&1)
tail -2 $O/prof_stdout.log | cut -c1-400
f=$(find $O/prof -name '*kernel_stats*' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0
for r in rows[:22]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.2f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f}")
print("sum of all kernels ms:", sum(float(r['TotalDurationNs']) for r in rows)/1e6)
PY
