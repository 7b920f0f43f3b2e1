#!/bin/bash
# GPU box, round 6: the phases of the common step (-DRB3_PROF_STEP) with 5, 2 and 1 waves per SIMD (unused LDS per block bounds the residency): what is latency, what is contention?
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
for lds in ${LDS_LIST:-0 64000 100000}; do
	RB3_EXP_DYNLDS=$lds RB3GPU_LIB=$R/ropebwt3_amd/prof/profstep.so timeout 900 python bench.py --only headline --no-aux --steps 1 --warmup 0 2>&1 >/dev/null | grep "prof\]" > gpurun_out/lonewave_$lds.log
	echo "== dynamic LDS $lds bytes per block"; for r in 20 100 150; do grep "common step x" gpurun_out/lonewave_$lds.log | sed -n "${r}p" | cut -c1-200; grep "general steps x" gpurun_out/lonewave_$lds.log | sed -n "${r}p" | cut -c1-330; done
done
grep "behind a flush" gpurun_out/lonewave_0.log | sed -n '20p;100p;150p'
