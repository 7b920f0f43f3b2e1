#!/bin/bash
# GPU box: kernel trace of `ropebwt3-amd build -L` on N reads (default 3 M), summary on stdout
R=${GRAFT_REPO_ROOT:-$PWD}; N=${1:-3000000}; export TMPDIR=/tmp
python $R/tools/gen_reads.py $N /tmp/reads.txt > /dev/null
cd /tmp; rm -rf /tmp/trace_reads
rocprofv3 --kernel-trace --stats -d /tmp/trace_reads -o rd -- $R/ropebwt3_amd/ropebwt3-amd build -L -d -m70m -o /tmp/o.fmd /tmp/reads.txt > /tmp/trace_reads.log 2>&1
cd $R
python tools/prof_summary.py stats $(ls /tmp/trace_reads/*_results.db /tmp/trace_reads/*/*_results.db 2>/dev/null | head -1) /tmp/rd_stats.txt | head -${2:-30} | cut -c1-150
