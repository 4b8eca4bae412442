#!/bin/bash
# GPU box: per-kernel event times of the default bench for the in-tree library and for variants/libsplat_<name>.so
#   tools/ab_bench.sh <kernel,kernel,...> [name ...]
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
K=$1; shift
run() { timeout 200 python bench.py --no-cpu-baseline < /dev/null 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); ks = '$K'.split(',')
print(d['value'], {k: d['kernels'][k]['avg_us'] for k in ks if k in d['kernels']})"; }
echo default; run
for v in "$@"; do echo $v; SPLAT_LIB_PATH=$GRAFT_REPO_ROOT/variants/libsplat_$v.so run; done
