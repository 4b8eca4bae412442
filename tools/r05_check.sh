#!/bin/bash
# GPU box: the round's new tests first (verbose), then whatever else is asked:  tools/r05_check.sh [full]
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r05_check
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train_step.py -x -q -s -m gpu < /dev/null > $O/train_step.log 2>&1; tail -25 $O/train_step.log
if [ "$1" = "full" ]; then
  timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_gpu_train_step.py < /dev/null > $O/pytest_gpu.log 2>&1; tail -15 $O/pytest_gpu.log
fi
