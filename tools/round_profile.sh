#!/bin/bash
# GPU box: everything the round's measurement record needs, in one gpurun call.
#   tools/round_profile.sh <tag>
cd $GRAFT_REPO_ROOT
TAG=${1:-r01f}
export PYTHONPATH=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q < /dev/null > gpurun_out/pytest_gpu_$TAG.log 2>&1; tail -3 gpurun_out/pytest_gpu_$TAG.log
timeout 300 python bench.py < /dev/null 2> /dev/null | tail -1 > gpurun_out/bench_$TAG.json; cut -c1-400 gpurun_out/bench_$TAG.json
timeout 300 python bench.py --ops --no-cpu-baseline < /dev/null 2> /dev/null | tail -1 > gpurun_out/bench_${TAG}_ops.json; cut -c1-200 gpurun_out/bench_${TAG}_ops.json
timeout 300 python bench.py --gaussians 1000000 --width 1280 --height 720 --no-cpu-baseline < /dev/null 2> /dev/null | tail -1 > gpurun_out/bench_${TAG}_c4.json; cut -c1-200 gpurun_out/bench_${TAG}_c4.json
timeout 300 python bench.py --channels 32 --no-cpu-baseline < /dev/null 2> /dev/null | tail -1 > gpurun_out/bench_${TAG}_c5.json; cut -c1-200 gpurun_out/bench_${TAG}_c5.json
timeout 300 python bench.py --dynamic --no-cpu-baseline < /dev/null 2> /dev/null | tail -1 > gpurun_out/bench_${TAG}_dynamic.json; cut -c1-200 gpurun_out/bench_${TAG}_dynamic.json
PYTHONPATH=. timeout 200 python tools/flow_bench.py < /dev/null 2> /dev/null | grep us > gpurun_out/flow_${TAG}.txt; cat gpurun_out/flow_${TAG}.txt
bash tools/prof_round.sh $TAG < /dev/null
B="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --frames 3 --no-cpu-baseline --no-kernel-timing"
bash tools/pmc_run.sh ${TAG}_rd "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum" $B < /dev/null > /dev/null
bash tools/pmc_run.sh ${TAG}_wr "WRITE_SIZE" $B < /dev/null > /dev/null
python tools/pmc_traffic.py gpurun_out/pmc_${TAG}_rd gpurun_out/pmc_${TAG}_wr gpurun_out/pmc_traffic_$TAG.json "rocprofv3 --pmc TCC_EA0_RDREQ_{32B,64B,128B}_sum (bytes = 32*n32+64*n64+128*n128) and WRITE_SIZE (KiB), separate passes, bench.py --frames 3 at configs[1], per launch" > /dev/null
BB="python $GRAFT_REPO_ROOT/tools/blend_bench.py --reps 2"
bash tools/pmc_run.sh ${TAG}_b1 "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA" $BB < /dev/null > /dev/null
bash tools/pmc_run.sh ${TAG}_b2 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS_F32 SQ_VALU_MFMA_BUSY_CYCLES" $BB < /dev/null > /dev/null
bash tools/pmc_run.sh ${TAG}_b3 "SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE" $BB < /dev/null > /dev/null
python tools/pmc_blend_counters.py gpurun_out/pmc_blend_counters_$TAG.json gpurun_out/pmc_${TAG}_b1 gpurun_out/pmc_${TAG}_b2 gpurun_out/pmc_${TAG}_b3
ls gpurun_out | grep $TAG
