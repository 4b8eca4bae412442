#!/bin/bash
# compaction of the forward's cull: parity suites that touch the quarter-list kernels, then the bench
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r06b
mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_frames_oracle.py tests/test_gpu_parity.py tests/test_gpu_consistency.py tests/test_gpu_determinism.py tests/test_gpu_block_kernels.py -x -q < /dev/null > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 600 python bench.py --no-cpu-baseline < /dev/null 2> $O/bench.err | tail -1 > $O/bench_line.json; python - <<'PY'
import json
d=json.load(open("gpurun_out/r06b/bench_line.json"))
print(d["summary"])
for k,v in d["kernels"].items(): print(k, v["us_per_frame"])
ts=d["extra_lines"][1]
print(ts.get("train_step_ms"), ts.get("phases_ms"))
for k,v in (ts.get("kernels_us_per_step") or {}).items(): print(k, v)
PY
tail -5 $O/bench.err
