cd $GRAFT_REPO_ROOT
timeout 200 python bench.py --no-cpu-baseline --dynamic < /dev/null 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_frame','gpu_kernel_ms_per_frame')}); print({k:(v['avg_us'],v['GBps']) for k,v in d['kernels'].items()}); print(d['config']['path'], d['config']['tile_pairs_M'])"
timeout 200 python bench.py --no-cpu-baseline < /dev/null 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_frame')})"
