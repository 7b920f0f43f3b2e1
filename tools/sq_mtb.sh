#!/bin/bash
# GPU box: SQ issue/stall counters (separate --pmc passes, kernel trace only) of `ropebwt3-amd build` on K genomes of the
# synthetic mtb star, averaged per dispatch for the kernels whose name contains PATTERN.   bash tools/sq_mtb.sh K PATTERN [tag]
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
K=${1:-100}; PAT=${2:-k_reb_group}; TAG=${3:-r2_sq_mtb$K}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/prof
python $R/tools/gen_mtb.py $K 4400000 /tmp/mtb_star_4400000 > /dev/null
FILES=$(ls /tmp/mtb_star_4400000/g*.fa | head -$K)
cd /tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"; do
	i=$((i+1))
	rocprofv3 --pmc $grp --kernel-trace -d $R/gpurun_out/prof/sq_$TAG.$i -o sq -- $R/ropebwt3_amd/ropebwt3-amd build -d -o /tmp/out.fmd $FILES > $R/gpurun_out/prof/sq_$TAG.$i.log 2>&1
done
cd $R
: > gpurun_out/prof/${TAG}.txt
for i in 1 2 3 4; do
	python - $(ls gpurun_out/prof/sq_$TAG.$i/*_results.db gpurun_out/prof/sq_$TAG.$i/*/*_results.db 2>/dev/null | head -1) "$PAT" >> gpurun_out/prof/${TAG}.txt <<'PY'
import sqlite3, sys
sys.path.insert(0, "tools")
from prof_summary import short
con = sqlite3.connect(sys.argv[1])
cols = [d[0] for d in con.execute("select * from counters_collection limit 1").description]
ncol = "counter_name" if "counter_name" in cols else "name"
kcol = "kernel_name" if "kernel_name" in cols else "name"
for k, c, n, a, mx in con.execute("select %s, %s, count(*), avg(value), max(value) from counters_collection group by %s, %s" % (kcol, ncol, kcol, ncol)):
    if sys.argv[2] in k: print("%-26s %-22s calls %4d avg %16.1f max %16.1f" % (short(k), c, n, a, mx))
PY
done
cat gpurun_out/prof/${TAG}.txt
rm -rf gpurun_out/prof/trace_$TAG gpurun_out/prof/sq_$TAG.* # the raw databases are large; gpurun copies at most 64 MiB back
