cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_fused.py -x -q < /dev/null 2>&1 | tail -6
timeout 200 python bench.py --no-cpu-baseline < /dev/null 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_frame')}); print({k:v['avg_us'] for k,v in d['kernels'].items()})"
timeout 200 python bench.py --no-cpu-baseline --gaussians 1000000 --width 1280 --height 720 < /dev/null 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_frame')}); print({k:v['avg_us'] for k,v in d['kernels'].items() if 'sort' in k or 'bin' in k})"
