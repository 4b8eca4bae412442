#!/bin/bash
# first GPU call of round 6: the new tests, then the default bench line
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r06a
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_shims.py tests/test_gpu_rccl.py tests/test_gpu_bench_line.py tests/test_gpu_knn.py tests/test_gpu_optim.py "tests/test_gpu_dp.py::test_densification_statistic_is_world_size_invariant" tests/test_gpu_train_step.py tests/test_gpu_arap.py -x -q < /dev/null > $O/pytest_new.log 2>&1; tail -15 $O/pytest_new.log
timeout 600 python bench.py < /dev/null 2> $O/bench.err | tail -1 > $O/bench_line.json; tail -c 1500 $O/bench_line.json; tail -5 $O/bench.err
