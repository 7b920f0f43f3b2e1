#!/bin/bash
# GPU box: what the settle kernels work on (stretch ids, walkers per merge), then library variants on the headline
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
RB3_BENCH_VERBOSE=4 timeout 600 python bench.py --only headline --steps 1 --warmup 0 2> gpurun_out/r5_settle_counts.err > /dev/null
grep "stretch ids" gpurun_out/r5_settle_counts.err | awk 'NR%10==1' | tail -16 > gpurun_out/r5_settle_counts.txt
cat gpurun_out/r5_settle_counts.txt
REPS=${REPS:-2} bash tools/gpu_ab_lib.sh "$@" 2>&1 | tee gpurun_out/r5_ab_settle.txt
