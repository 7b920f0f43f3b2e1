#!/bin/bash
# where does k_reb_group spend its cycles on the headline workload?  (RB3_PROF_REB build)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
for k in 40 152; do
	echo "== mtb $k"
	RB3GPU_LIB=$R/ropebwt3_amd/prof/profreb.so timeout 300 python bench.py --only headline --steps 1 --warmup 0 --mtb $k 2>&1 >/dev/null | grep "prof\]"
done
