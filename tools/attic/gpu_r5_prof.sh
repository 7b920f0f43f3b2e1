#!/bin/bash
# round 5: phase timers of the common step (RB3_PROF_STEP build)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
export RB3GPU_LIB=$R/ropebwt3_amd/prof/${1:-profstep}.so
timeout 300 python bench.py --only headline --steps 1 --warmup 0 2> gpurun_out/r5_profstep.err > /dev/null
grep "common step" gpurun_out/r5_profstep.err | awk 'NR%30==1' | head -6
grep "common step" gpurun_out/r5_profstep.err | tail -1
