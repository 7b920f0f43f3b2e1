#!/bin/bash
# GPU box, round 6: kernel statistics (rocprofv3 --kernel-trace --stats) of ONE headline build per library variant -- for variants whose results are
# wrong on purpose (stores taken out: the merge is redone by the non-tentative kernel, whose launches have a row of their own) the per-kernel average is the number
#   bash tools/r6/exp3_kstats.sh name1 name2 ...      ("release" = the in-tree library)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out/prof
for v in "$@"; do
	if [ "$v" = release ]; then unset RB3GPU_LIB; else export RB3GPU_LIB=$R/ropebwt3_amd/prof/$v.so; fi
	LINES_OUT=60 bash tools/prof_bench.sh x_$v --only headline --no-aux --steps 1 --warmup 0 > gpurun_out/prof/x_$v.txt 2>&1
	echo "== $v"; grep -i "k_chain\|k_events\|k_cum\|TOTAL" gpurun_out/prof/x_$v.txt | cut -c1-150
done
