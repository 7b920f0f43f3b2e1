#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
export RB3GPU_LIB=$R/ropebwt3_amd/prof/profstep.so
timeout 300 python bench.py --only headline --steps 1 --warmup 0 2> gpurun_out/exp6.err > /dev/null
grep "common step" gpurun_out/exp6.err | awk 'NR%15==1' | cut -c 50-400
