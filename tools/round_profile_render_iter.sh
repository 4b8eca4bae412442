#!/bin/bash
# GPU box: kernel trace and issue counters of the reference's training frame (bench.py --render-iter --dynamic)
#   tools/round_profile_render_iter.sh <tag>
cd $GRAFT_REPO_ROOT
TAG=${1:-r05}
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
O=gpurun_out/${TAG}_ri
mkdir -p $O
B="python $GRAFT_REPO_ROOT/bench.py --render-iter --dynamic --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing"
(cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o bench -- $B > $GRAFT_REPO_ROOT/$O/prof_stdout.log 2>&1)
python tools/trace_workload_stats.py $(find $O/prof -name '*kernel_trace.csv' | head -1) $O/kernel_stats_workload.csv
B1="python $GRAFT_REPO_ROOT/bench.py --render-iter --dynamic --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing"
bash tools/pmc_run.sh ${TAG}ri_b1 "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA" $B1 < /dev/null > /dev/null
bash tools/pmc_run.sh ${TAG}ri_b2 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS_F32 SQ_VALU_MFMA_BUSY_CYCLES" $B1 < /dev/null > /dev/null
bash tools/pmc_run.sh ${TAG}ri_b3 "SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE" $B1 < /dev/null > /dev/null
PMC_CONFIG="300000x854x480x0:render_iter:dynamic:morton" PMC_SOURCE="rocprofv3 --pmc, three passes over one step of bench.py --render-iter --dynamic (25 frames per launch), per launch, summed over the 8 XCDs; quad-cycle counters as in r02_pmc_blend_counters.json" python tools/pmc_blend_counters.py $O/pmc_blend_counters.json gpurun_out/pmc_${TAG}ri_b1 gpurun_out/pmc_${TAG}ri_b2 gpurun_out/pmc_${TAG}ri_b3
head -6 $O/kernel_stats_workload.csv
