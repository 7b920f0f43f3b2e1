#!/bin/bash
# GPU box: how old a walker must be to record tentatively (RB3_TENT_MIN_AGE = the pre-roll of the device-made list): headline and crowded hunt per variant
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
REPS=1 bash tools/gpu_ab_lib.sh release age16 age12 age8 2>&1 | tee gpurun_out/r5_ab_age.txt
for v in age16 age12 age8; do
	echo "== crowded hunt, $v" | tee -a gpurun_out/r5_ab_age.txt
	RB3GPU_LIB=$R/ropebwt3_amd/prof/$v.so RB3GPU_JUNCTION_CHECK=1 bash tools/gpu_crowded_hunt.sh ${RUNS:-6} 40 2>&1 | grep -v "^\[W" | tail -3 | tee -a gpurun_out/r5_ab_age.txt
done
