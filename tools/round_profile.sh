#!/bin/bash
# GPU box: everything the round's measurement record needs, in one gpurun call.
#   tools/round_profile.sh <tag>        (writes gpurun_out/<tag>/...)
cd $GRAFT_REPO_ROOT
TAG=${1:-r05}
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/$TAG
mkdir -p $O
python - <<PY > $O/stamp.json
import json, sys
sys.path.insert(0, "tools")
from stamp import stamp
print(json.dumps(stamp(), indent=1))
PY
cat $O/stamp.json
run() { name=$1; shift; timeout 400 python bench.py "$@" < /dev/null 2> $O/$name.err | tail -1 > $O/$name.json; cut -c1-160 $O/$name.json; }
timeout 900 python -m pytest tests -m gpu -q < /dev/null > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
run bench_line                                   # default: frame batch, synchronous step with Adam, CPU baselines
run bench_random_order --no-spatial-order --no-cpu-baseline
run bench_per_frame --per-frame --no-cpu-baseline
run bench_operator_chain --ops --no-cpu-baseline
run bench_c4_1M_720p --gaussians 1000000 --width 1280 --height 720 --no-cpu-baseline
run bench_c5_32ch --channels 32 --no-cpu-baseline
run bench_dynamic_a15 --dynamic --no-cpu-baseline
run bench_render_iter --render-iter --no-cpu-baseline
run bench_render_iter_per_frame --render-iter --per-frame --no-cpu-baseline
run bench_render_iter_dynamic --render-iter --dynamic --no-cpu-baseline   # the reference's real training frame
run bench_train_step --train-step --no-cpu-baseline                         # the composed training step (train_step.py)
run bench_clustered --scene clustered --no-cpu-baseline --no-extra-lines   # 70 % of the Gaussians in blobs covering 10 % of the image
run bench_ref_flow --ref-flow --steps 3 --warmup 1 --no-cpu-baseline      # the reference's literal render_iter call sequence (eager projection + EWA)
# kernel trace of the default bench command (2 timed steps of 25 frames)
export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-extra-lines"
(cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o bench -- $B > $GRAFT_REPO_ROOT/$O/prof_stdout.log 2>&1)
find $O/prof -name '*kernel_stats*' | head -3
python tools/trace_workload_stats.py $(find $O/prof -name '*kernel_trace.csv' | head -1) $O/kernel_stats_workload.csv
# HBM traffic: two --pmc passes over one step of the default bench (25 frames per launch)
B1="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-extra-lines"
bash tools/pmc_run.sh ${TAG}_rd "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum" $B1 < /dev/null > /dev/null
bash tools/pmc_run.sh ${TAG}_wr "WRITE_SIZE" $B1 < /dev/null > /dev/null
python tools/pmc_traffic.py gpurun_out/pmc_${TAG}_rd gpurun_out/pmc_${TAG}_wr $O/pmc_traffic.json "rocprofv3 --pmc TCC_EA0_RDREQ_{32B,64B,128B}_sum (bytes = 32*n32+64*n64+128*n128) and WRITE_SIZE (KiB), separate passes, one step of the default bench.py (frame batch: 25 frames per launch), per launch" "300000x854x480x0:batch:morton" > /dev/null
# issue counters of the two compositing kernels on the same command (three passes)
bash tools/pmc_run.sh ${TAG}_b1 "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA" $B1 < /dev/null > /dev/null
bash tools/pmc_run.sh ${TAG}_b2 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS_F32 SQ_VALU_MFMA_BUSY_CYCLES" $B1 < /dev/null > /dev/null
bash tools/pmc_run.sh ${TAG}_b3 "SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE" $B1 < /dev/null > /dev/null
PMC_CONFIG="300000x854x480x0:batch:morton" PMC_SOURCE="rocprofv3 --pmc, three passes over one step of the default bench.py (frame batch: 25 frames per launch), per launch, summed over the 8 XCDs; SQ_ACTIVE_INST_* / SQ_WAIT_* / SQ_WAVE_CYCLES in quad-cycles (guides/MI355X_MICROARCH.md)" python tools/pmc_blend_counters.py $O/pmc_blend_counters.json gpurun_out/pmc_${TAG}_b1 gpurun_out/pmc_${TAG}_b2 gpurun_out/pmc_${TAG}_b3
# clustered scene (70 % of the Gaussians on 10 % of the image): wave residency of the compositing kernels (two passes)
BC="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --scene clustered --no-cpu-baseline --no-kernel-timing --no-extra-lines"
bash tools/pmc_run.sh ${TAG}_c1 "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA" $BC < /dev/null > /dev/null
bash tools/pmc_run.sh ${TAG}_c3 "SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE" $BC < /dev/null > /dev/null
PMC_CONFIG="300000x854x480x0:batch:morton:clustered" PMC_SOURCE="rocprofv3 --pmc, two passes over one step of bench.py --scene clustered, per launch, summed over the 8 XCDs" python tools/pmc_blend_counters.py $O/pmc_blend_counters_clustered.json gpurun_out/pmc_${TAG}_c1 gpurun_out/pmc_${TAG}_c3
# the default line once more, now that the counter record of THIS build exists: roofline.traffic / roofline.issue are quoted only
# from a record whose build id is the running library's (bench.py::pmc_stamp_ok)
cp $O/pmc_traffic.json profiles/${TAG}_pmc_traffic.json; cp $O/pmc_blend_counters.json profiles/${TAG}_pmc_blend_counters.json
run bench_line_pmc
bash tools/round_profile_render_iter.sh $TAG > $O/render_iter_profile.log 2>&1; tail -4 $O/render_iter_profile.log
ls $O
