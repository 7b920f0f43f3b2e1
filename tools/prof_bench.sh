#!/bin/bash
# GPU box: rocprofv3 kernel trace of a bench.py invocation; summary -> gpurun_out/prof/<tag>_kernel_stats.txt
# bash tools/prof_bench.sh TAG [bench args...]
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r2}; shift
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/prof
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/trace_$TAG -o $TAG -- python $R/bench.py "$@" > $R/gpurun_out/prof/trace_$TAG.log 2>&1
cd $R
python tools/prof_summary.py stats $(ls gpurun_out/prof/trace_$TAG/*_results.db gpurun_out/prof/trace_$TAG/*/*_results.db 2>/dev/null | head -1) gpurun_out/prof/${TAG}_kernel_stats.txt | head -${LINES_OUT:-30}
grep -h '"metric"\|^{"workload"' gpurun_out/prof/trace_$TAG.log > gpurun_out/prof/${TAG}_bench_under_rocprof.json; [ -s gpurun_out/prof/${TAG}_bench_under_rocprof.json ] || rm -f gpurun_out/prof/${TAG}_bench_under_rocprof.json
rm -rf gpurun_out/prof/trace_$TAG
