#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats of the default bench command.
# usage: tools/prof_round.sh <tag>
set -u
TAG=${1:-r01}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --frames 5 --no-cpu-baseline > $OUT/bench_stdout.log 2>&1
cd $GRAFT_REPO_ROOT
find $OUT -name '*kernel_stats*' | head
