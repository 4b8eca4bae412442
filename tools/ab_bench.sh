#!/bin/bash
# GPU box: per-kernel event times of the default bench for the in-tree library and for variants/libsplat_<name>.so
#   tools/ab_bench.sh <kernel,kernel,...> [name ...]          (BENCH_ARGS: extra bench.py flags, ROUNDS: interleaved repeats)
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
K=$1; shift
run() { timeout 300 python bench.py --no-cpu-baseline --no-extra-lines --steps 6 $BENCH_ARGS < /dev/null 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); ks = '$K'.split(',')
print(d['value'], {k: d['kernels'][k]['us_per_frame'] for k in ks if k in d['kernels']})"; }
for r in $(seq 1 ${ROUNDS:-2}); do
echo "round $r: default"; run
for v in "$@"; do echo "round $r: $v"; SPLAT_LIB_PATH=$GRAFT_REPO_ROOT/variants/libsplat_$v.so run; done
done
