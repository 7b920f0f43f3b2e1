#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace + separate PMC passes of bench.py, summaries written to
# gpurun_out/prof/<tag>_* (copy the ones to keep into profiles/).  PMC passes use --kernel-trace only (never --sys-trace etc.
# together with --pmc).   bash tools/run_profiles.sh TAG KERNEL-SUBSTRING OUT-NAME [bench args...]
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r2}; KPAT=${2:-k_chain}; KOUT=${3:-k_chain}; shift 3
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/prof
cd /tmp
BENCH="python $R/bench.py --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/trace -o $TAG -- $BENCH > $R/gpurun_out/prof/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/prof/pmc_fetch -o $TAG -- $BENCH > $R/gpurun_out/prof/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/prof/pmc_write -o $TAG -- $BENCH > $R/gpurun_out/prof/pmc_write.log 2>&1
cd $R
P=gpurun_out/prof
db() { ls $P/$1/*_results.db $P/$1/*/*_results.db 2>/dev/null | head -1; }
python tools/prof_summary.py stats $(db trace) $P/${TAG}_kernel_stats.txt > /dev/null
python tools/prof_summary.py pmc $(db pmc_fetch) FETCH_SIZE $P/${TAG}_pmc_fetch_size.txt > /dev/null
python tools/prof_summary.py pmc $(db pmc_write) WRITE_SIZE $P/${TAG}_pmc_write_size.txt > /dev/null
python tools/prof_summary.py traffic $(db pmc_fetch) $(db pmc_write) "$KPAT" $P/${TAG}_pmc_${KOUT}.json > /dev/null
grep -h '"metric"' $P/trace.log > $P/${TAG}_bench_under_rocprof.json
head -14 $P/${TAG}_kernel_stats.txt | cut -c1-140
head -5 $P/${TAG}_pmc_fetch_size.txt | cut -c1-120; head -5 $P/${TAG}_pmc_write_size.txt | cut -c1-120
cat $P/${TAG}_pmc_${KOUT}.json | head -12
rm -rf $P/trace $P/pmc_fetch $P/pmc_write
