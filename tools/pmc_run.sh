#!/bin/bash
# usage: tools/pmc_run.sh <tag> "<counters>" <cmd...>   (GPU box; counters in their own pass)
TAG=$1; shift; CNT=$1; shift
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp
timeout -k 5 240 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT -o pmc -- "$@" > $OUT/stdout.log 2>&1
echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
ls $OUT
