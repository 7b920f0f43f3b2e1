#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/tl; rocprofv3 --kernel-trace -d /tmp/tl -o tl -- python $R/bench.py --only headline --steps 1 --warmup 0 --mtb ${K:-100} > /tmp/tl.log 2>&1
cd $R; python tools/timeline.py $(ls /tmp/tl/*_results.db /tmp/tl/*/*_results.db 2>/dev/null | head -1) ${ROUND:-}
