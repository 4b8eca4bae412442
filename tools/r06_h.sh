#!/bin/bash
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_frames_oracle.py tests/test_gpu_frames.py tests/test_gpu_renderer_native.py -x -q < /dev/null 2>&1 | tail -2
for a in 4 1; do
timeout 600 python bench.py --render-iter --attr-channels $a --no-cpu-baseline --no-extra-lines --no-other-configs < /dev/null 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], {k: d['kernels'][k]['us_per_frame'] for k in ('blend_fwd','blend_bwd','gauss_bwd','blend_pack') if k in d['kernels']})"
done
