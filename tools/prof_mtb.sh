#!/bin/bash
# GPU box: rocprofv3 kernel trace of `ropebwt3-amd build` on K genomes of the synthetic mtb star; summary in gpurun_out/prof/<tag>_kernel_stats.txt
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
K=${1:-100}; TAG=${2:-r2_mtb$K}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/prof
python $R/tools/gen_mtb.py $K 4400000 /tmp/mtb_star_4400000 > /dev/null
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/trace_$TAG -o $TAG -- $R/ropebwt3_amd/ropebwt3-amd build -d -o /tmp/out.fmd $(ls /tmp/mtb_star_4400000/g*.fa | head -$K) > $R/gpurun_out/prof/trace_$TAG.log 2>&1
cd $R
python tools/prof_summary.py stats $(ls gpurun_out/prof/trace_$TAG/*/*_results.db gpurun_out/prof/trace_$TAG/*_results.db 2>/dev/null | head -1) gpurun_out/prof/${TAG}_kernel_stats.txt | head -40
grep "GPU merge path" gpurun_out/prof/trace_$TAG.log
rm -rf gpurun_out/prof/trace_$TAG gpurun_out/prof/sq_$TAG.* # the raw databases are large; gpurun copies at most 64 MiB back
