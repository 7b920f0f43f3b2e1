#!/bin/bash
# GPU box, round 6: where does a common step of k_chain go now?  (s_memtime at four points, lane 0 of every wave; -DRB3_PROF_STEP variant)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
RB3GPU_LIB=$R/ropebwt3_amd/prof/profstep.so timeout 600 python bench.py --only headline --no-aux --steps 1 --warmup 0 2>&1 >/dev/null | grep "prof\]" > gpurun_out/profstep.log
wc -l gpurun_out/profstep.log
for r in 5 20 50 100 150 170 250 300; do sed -n "${r}p" gpurun_out/profstep.log | sed "s/^/merge $r: /" | cut -c1-330; done
