#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
for v in profstep ps_norec ps_noev ps_nost ps_plainrec; do
export RB3GPU_LIB=$R/ropebwt3_amd/prof/$v.so
timeout 600 python bench.py --only headline --steps 1 --warmup 0 --mtb 60 2> gpurun_out/exp7.err > /dev/null
echo "== $v"; grep "common step" gpurun_out/exp7.err | awk 'NR==20 || NR==55' | cut -c 50-200
done
