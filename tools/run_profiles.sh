#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace + separate PMC passes of the default
# bench command, summaries written to gpurun_out/prof/*.txt|json (copy the ones to keep into profiles/).
# PMC passes use --kernel-trace only (never --sys-trace etc. together with --pmc).
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r1}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/prof
cd /tmp
BENCH="python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-aux"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/trace -o $TAG -- $BENCH > $R/gpurun_out/prof/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/prof/pmc_fetch -o $TAG -- $BENCH > $R/gpurun_out/prof/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/prof/pmc_write -o $TAG -- $BENCH > $R/gpurun_out/prof/pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $R/gpurun_out/prof/pmc_l2 -o $TAG -- $BENCH > $R/gpurun_out/prof/pmc_l2.log 2>&1
cd $R
P=gpurun_out/prof
python tools/prof_summary.py stats $P/trace/${TAG}_results.db $P/${TAG}_kernel_stats.txt > /dev/null
python tools/prof_summary.py pmc $P/pmc_fetch/${TAG}_results.db FETCH_SIZE $P/${TAG}_pmc_fetch_size.txt > /dev/null
python tools/prof_summary.py pmc $P/pmc_write/${TAG}_results.db WRITE_SIZE $P/${TAG}_pmc_write_size.txt > /dev/null
python tools/prof_summary.py pmc $P/pmc_l2/${TAG}_results.db TCC_HIT_sum $P/${TAG}_pmc_tcc_hit.txt > /dev/null
python tools/prof_summary.py pmc $P/pmc_l2/${TAG}_results.db TCC_MISS_sum $P/${TAG}_pmc_tcc_miss.txt > /dev/null
python tools/prof_summary.py traffic $P/pmc_fetch/${TAG}_results.db $P/pmc_write/${TAG}_results.db k_chain $P/${TAG}_pmc_k_chain.json > /dev/null
grep -h '"metric"' $P/trace.log > $P/${TAG}_bench_under_rocprof.json
cat $P/${TAG}_kernel_stats.txt | head -8
head -4 $P/${TAG}_pmc_fetch_size.txt; head -4 $P/${TAG}_pmc_write_size.txt; head -3 $P/${TAG}_pmc_tcc_hit.txt; head -3 $P/${TAG}_pmc_tcc_miss.txt
