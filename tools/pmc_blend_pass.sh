#!/bin/bash
# GPU box: the three --pmc passes over tools/blend_bench.py (single-frame launches) for a channel count
#   tools/pmc_blend_pass.sh <tag> [channels]
cd $GRAFT_REPO_ROOT
TAG=${1:-r02q}; CH=${2:-3}
BB="python $GRAFT_REPO_ROOT/tools/blend_bench.py --reps 2 --channels $CH"
bash tools/pmc_run.sh ${TAG}_b1 "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA" $BB < /dev/null > /dev/null
bash tools/pmc_run.sh ${TAG}_b2 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS_F32 SQ_VALU_MFMA_BUSY_CYCLES" $BB < /dev/null > /dev/null
bash tools/pmc_run.sh ${TAG}_b3 "SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE" $BB < /dev/null > /dev/null
python tools/pmc_blend_counters.py gpurun_out/pmc_blend_counters_$TAG.json gpurun_out/pmc_${TAG}_b1 gpurun_out/pmc_${TAG}_b2 gpurun_out/pmc_${TAG}_b3
