#!/bin/bash
# three rounds of the 152-genome build as timelines (late, middle: mixed groups, early: bit planes)
R=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/tl; rocprofv3 --kernel-trace -d /tmp/tl -o tl -- python $R/bench.py --only headline --steps 1 --warmup 0 --mtb 152 > /tmp/tl.log 2>&1
cd $R; DB=$(ls /tmp/tl/*_results.db /tmp/tl/*/*_results.db 2>/dev/null | head -1)
for r in ${ROUNDS:-8 30 60 100 145}; do echo "=== round $r"; python tools/timeline.py $DB $r $RAW; done
