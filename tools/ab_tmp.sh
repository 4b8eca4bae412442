cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
run() { timeout 100 python tools/blend_bench.py "$@" < /dev/null 2>&1 | grep "^\[" ; }
echo base; run
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_renderer_flow.py tests/test_gpu_golden.py -x -q < /dev/null 2>&1 | tail -4
