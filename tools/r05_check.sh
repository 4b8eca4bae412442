#!/bin/bash
# GPU box: the round's new tests first (verbose), then whatever else is asked:  tools/r05_check.sh [full] [bench]
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r05_check
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train_step.py -x -q -m gpu < /dev/null > $O/train_step.log 2>&1; tail -5 $O/train_step.log
timeout 1800 python -m pytest tests/test_gpu_dp.py -x -q -m gpu < /dev/null > $O/dp.log 2>&1; tail -25 $O/dp.log
for a in "$@"; do
  if [ "$a" = "full" ]; then
    timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_gpu_train_step.py --deselect tests/test_gpu_dp.py < /dev/null > $O/pytest_gpu.log 2>&1; tail -15 $O/pytest_gpu.log
  fi
  if [ "$a" = "bench" ]; then
    timeout 600 python bench.py < /dev/null 2> $O/bench.err | tail -1 > $O/bench.json; python - <<PY
import json
d = json.load(open("$O/bench.json"))
print("value", d["value"], "ms/frame", d["ms_per_frame"], "fwd_only", d["forward_only"]["value"])
print("roofline", {k: d["roofline"][k] for k in ("kernel", "bound", "limited_by", "frac", "frac_hbm", "frac_fp32_issue")})
print({k: v["us_per_frame"] for k, v in d["kernels"].items()})
for e in d["extra_lines"]:
    print({k: e[k] for k in e if k not in ("config", "metric", "reference_context")})
PY
    tail -3 $O/bench.err
  fi
done
