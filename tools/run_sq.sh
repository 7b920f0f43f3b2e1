#!/bin/bash
# SQ issue/stall counters of the default bench command (one --pmc pass per counter group, kernel trace only).
# Shows whether k_chain is bound by VALU issue or by waiting for memory.  Output: gpurun_out/prof/<tag>_pmc_sq.txt
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r1}
EXTRA=${2:-}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/prof
cd /tmp
BENCH="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline $EXTRA"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM"; do
	i=$((i+1))
	rocprofv3 --pmc $grp --kernel-trace -d $R/gpurun_out/prof/pmc_sq$i -o $TAG -- $BENCH > $R/gpurun_out/prof/pmc_sq$i.log 2>&1
done
cd $R
P=gpurun_out/prof
: > $P/${TAG}_pmc_sq.txt
for i in 1 2 3 4; do
	python - $P/pmc_sq$i/${TAG}_results.db >> $P/${TAG}_pmc_sq.txt <<'PY'
import sqlite3, sys
sys.path.insert(0, "tools")
from prof_summary import short
try:
    con = sqlite3.connect(sys.argv[1])
    cols = [d[0] for d in con.execute("select * from counters_collection limit 1").description]
    ncol = "counter_name" if "counter_name" in cols else "name"
    kcol = "kernel_name" if "kernel_name" in cols else "name"
    rows = con.execute("select %s, %s, count(*), avg(value) from counters_collection group by %s, %s" % (kcol, ncol, kcol, ncol)).fetchall()
    for k, c, n, a in rows:
        if "k_chain" in k or "k_pass" in k:
            print("%-28s %-24s calls %6d avg_per_dispatch %16.1f" % (short(k), c, n, a))
except Exception as e:
    print("ERR", sys.argv[1], e)
PY
done
cat $P/${TAG}_pmc_sq.txt
