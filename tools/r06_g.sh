#!/bin/bash
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r06g
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_knn.py tests/test_gpu_train_step.py -x -q < /dev/null 2>&1 | tail -2
timeout 600 python bench.py --train-step --no-cpu-baseline < /dev/null 2> $O/bench.err | tail -1 > $O/bench_ts.json; python - <<'PY'
import json
ts=json.load(open("gpurun_out/r06g/bench_ts.json"))
print(ts.get("train_step_ms"), ts.get("phases_ms"))
k=ts.get("kernels_us_per_step") or {}
print({n: k[n]["us_per_step"] for n in k if n.startswith(("knn","arap","blend","gauss"))})
PY
