#!/bin/bash
# where does a step of k_chain go?  (RB3_PROF_STEP build: s_memtime at four points of the common step)   K=152 by default
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
K=${K:-152}
RB3GPU_LIB=$R/ropebwt3_amd/prof/profstep.so timeout 600 python bench.py --only headline --steps 1 --warmup 0 --mtb $K 2>&1 >/dev/null | grep "prof\]" > /tmp/p.log
wc -l /tmp/p.log
for r in ${ROUNDS:-20 50 100 150}; do sed -n "${r}p" /tmp/p.log | sed "s/^/round $r: /"; done
