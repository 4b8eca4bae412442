#!/bin/bash
# copies the round's measurement record from gpurun_out/<tag>/ (scratch) into profiles/<tag>_* (tracked)
TAG=${1:-r05}
S=gpurun_out/$TAG; R=gpurun_out/${TAG}_ri
for f in $S/bench_*.json; do b=$(basename $f); cp $f profiles/${TAG}_$b; done
cp $S/pmc_traffic.json profiles/${TAG}_pmc_traffic.json
cp $S/pmc_blend_counters.json profiles/${TAG}_pmc_blend_counters.json
cp $S/pmc_blend_counters_clustered.json profiles/${TAG}_pmc_blend_counters_clustered.json
cp $S/kernel_stats_workload.csv profiles/${TAG}_kernel_stats_workload.csv
cp $S/stamp.json profiles/${TAG}_stamp.json
find $S/prof -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} profiles/${TAG}_kernel_stats.csv
tail -40 $S/pytest_gpu.log > profiles/${TAG}_pytest_gpu.txt
cp $R/pmc_blend_counters.json profiles/${TAG}_pmc_blend_counters_training_frame.json
cp $R/kernel_stats_workload.csv profiles/${TAG}_kernel_stats_workload_training_frame.csv
ls profiles | grep "^${TAG}_" | wc -l
