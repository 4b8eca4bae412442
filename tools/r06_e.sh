#!/bin/bash
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r06e
mkdir -p $O
timeout 600 python bench.py --train-step --no-cpu-baseline --unfused-l1 < /dev/null 2> $O/bench.err | tail -1 > $O/bench_ts.json; python - <<'PY'
import json
ts=json.load(open("gpurun_out/r06e/bench_ts.json"))
print(ts.get("train_step_ms"), ts.get("phases_ms"))
k=ts.get("kernels_us_per_step") or {}
print({n: k[n]["us_per_step"] for n in ("blend_fwd","blend_bwd","gauss_bwd","l1_loss_grad") if n in k})
PY
