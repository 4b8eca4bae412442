#!/bin/bash
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r06c
mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_rccl.py tests/test_gpu_frames_oracle.py tests/test_gpu_frames.py tests/test_gpu_determinism.py tests/test_gpu_dp.py tests/test_gpu_block_kernels.py tests/test_gpu_bench_line.py -x -q < /dev/null > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 600 python bench.py --train-step --no-cpu-baseline < /dev/null 2> $O/bench.err | tail -1 > $O/bench_ts.json; python - <<'PY'
import json
ts=json.load(open("gpurun_out/r06c/bench_ts.json"))
print(ts.get("train_step_ms"), ts.get("phases_ms"))
for k,v in (ts.get("kernels_us_per_step") or {}).items(): print(k, v)
PY
tail -3 $O/bench.err
