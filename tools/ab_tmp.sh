cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
run() { timeout 100 python tools/blend_bench.py "$@" < /dev/null 2>&1 | grep "^\[" ; }
echo run16; run
for v in run8 run32 run54 run1; do echo $v; SPLAT_LIB_PATH=$GRAFT_REPO_ROOT/variants/libsplat_$v.so run; done
B="python $GRAFT_REPO_ROOT/tools/blend_bench.py --reps 2"
bash tools/pmc_run.sh n2 "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum" $B < /dev/null > /dev/null
