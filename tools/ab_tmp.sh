cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py tests/test_gpu_golden.py -x -q -k "sh or chain or golden or fused" < /dev/null 2>&1 | tail -8
timeout 200 python bench.py --no-cpu-baseline < /dev/null 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_frame','gpu_kernel_ms_per_frame')}); print({k:v['avg_us'] for k,v in d['kernels'].items()})"
