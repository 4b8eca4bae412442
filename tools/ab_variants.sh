#!/bin/bash
# GPU box: A/B of library variants (tools/build_variant.sh) on one bench command; prints frames/s and the two compositing kernels' us per frame
#   tools/ab_variants.sh "<bench flags>" default name1 name2 ...     (ROUNDS=2: repetitions, interleaved)
cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
flags=$1; shift
for r in $(seq 1 ${ROUNDS:-2}); do
  for v in "$@"; do
    if [ "$v" = default ]; then unset SPLAT_LIB_PATH; else export SPLAT_LIB_PATH=$GRAFT_REPO_ROOT/variants/libsplat_$v.so; fi
    python bench.py $flags --no-cpu-baseline --no-extra-lines 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d.get('kernels') or {}
g=lambda n: (k.get(n) or {}).get('us_per_frame')
print('$v', d['value'], 'fwd', g('blend_fwd'), 'bwd', g('blend_bwd'), 'sort', g('tile_sort'), 'gauss', g('gauss_bwd'))"
  done
done
