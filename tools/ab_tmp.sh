cd $GRAFT_REPO_ROOT
export PYTHONPATH=.
run() { timeout 100 python tools/blend_bench.py "$@" < /dev/null 2>&1 | grep "^\[" ; }
echo base; run
echo dpp; SPLAT_BWD_KERNEL=dpp run
for v in w2 w4 sb256 sb64; do echo $v; SPLAT_LIB_PATH=$GRAFT_REPO_ROOT/variants/libsplat_$v.so run; done
