#!/bin/bash
# GPU box: hardware counters of the headline workload (bench.py --only headline, one build), separate --pmc passes with
# --kernel-trace only; per-kernel averages of every counter -> gpurun_out/prof/<tag>_counters.txt, and FETCH/WRITE traffic of the
# kernels that match PATTERN -> <tag>_pmc_<name>.json.   bash tools/pmc_headline.sh TAG PATTERN NAME "GROUP1" "GROUP2" ...
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r3_mtb152}; PAT=${2:-k_chain}; NAME=${3:-k_chain_mtb152}; shift 3
export TMPDIR=/tmp
P=$R/gpurun_out/prof
mkdir -p $P
cd /tmp
BENCH="python $R/bench.py --only ${LEG:-headline} ${BENCH_STEPS:---steps 1 --warmup 0} ${BENCH_ARGS:-}"
i=0
: > $P/${TAG}_counters.txt
for grp in "$@"; do
	i=$((i+1))
	rm -rf $P/pmc_$TAG.$i
	timeout 600 rocprofv3 --pmc $grp --kernel-trace -d $P/pmc_$TAG.$i -o pmc -- $BENCH > $P/pmc_$TAG.$i.log 2>&1
	DB=$(ls $P/pmc_$TAG.$i/*_results.db $P/pmc_$TAG.$i/*/*_results.db 2>/dev/null | head -1)
	python - "$DB" "$PAT" >> $P/${TAG}_counters.txt <<'PY'
import sqlite3, sys
sys.path.insert(0, "/root/repo/tools")
from prof_summary import short
con = sqlite3.connect(sys.argv[1])
cols = [d[0] for d in con.execute("select * from counters_collection limit 1").description]
ncol = "counter_name" if "counter_name" in cols else "name"
kcol = "kernel_name" if "kernel_name" in cols else "name"
for k, c, n, a, mx in con.execute("select %s, %s, count(*), avg(value), max(value) from counters_collection group by %s, %s" % (kcol, ncol, kcol, ncol)):
    if any(p in k for p in sys.argv[2].split("|")): print("%-28s %-26s calls %5d avg %18.1f max %18.1f" % (short(k), c, n, a, mx))
PY
	case "$grp" in *FETCH_SIZE*) FDB=$DB;; esac
	case "$grp" in *WRITE_SIZE*) WDB=$DB;; esac
done
cd $R
if [ -n "${FDB:-}" ] && [ -n "${WDB:-}" ]; then python tools/prof_summary.py traffic $FDB $WDB "$PAT" $P/${TAG}_pmc_${NAME}.json; fi
cat $P/${TAG}_counters.txt
rm -rf $P/pmc_$TAG.*
