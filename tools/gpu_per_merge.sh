#!/bin/bash
# GPU box: the 151 merges of the headline build one by one (verbose 4: the engine prints every merge's phases from its HIP events)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
RB3_BENCH_VERBOSE=4 timeout 600 python bench.py --only headline --no-aux --steps 1 --warmup 1 2> gpurun_out/r5_series.err > /dev/null
grep "fill-to-walkers" gpurun_out/r5_series.err | tail -151 | awk '{ n++; printf "%d into %s steps %s ids %s: fill %s k_chain %s settle %s rebuild %s\n", n, $7, $12, $17, $25, $28, $32, $34 }' | sed 's/[,:;]//g' > gpurun_out/r5_series.txt
awk 'NR%10==1 || NR==151' gpurun_out/r5_series.txt
