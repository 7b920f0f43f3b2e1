#!/bin/bash
# GPU box, round 6: the kernels of ONE merge late in the 152-genome build in the order they ran -- start (us from the merge's first kernel), duration, the gap in
# front of each on its queue -- from a rocprofv3 kernel trace of the headline leg.  Which microseconds of a merge is the chip not working?
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out/prof
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/prof/trace_tl -o tl -- python $R/bench.py --only headline --steps 1 --warmup 0 > $R/gpurun_out/prof/trace_tl.json 2> $R/gpurun_out/prof/trace_tl.err
cd $R
DB=$(ls gpurun_out/prof/trace_tl/*_results.db gpurun_out/prof/trace_tl/*/*_results.db 2>/dev/null | head -1)
python - "$DB" ${1:-140} <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); which = int(sys.argv[2])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in db.execute("pragma table_info(%s)" % kd)]
q = "queue_id" if "queue_id" in cols else "0"
rows = db.execute("select s.kernel_name, d.start, d.end, d.%s from %s d join %s s on d.kernel_id = s.id order by d.start" % (q, kd, ks)).fetchall()
# merges begin with k_fill_regions
starts = [i for i, r in enumerate(rows) if "k_fill_regions" in r[0]]
print("%d dispatches, %d merges" % (len(rows), len(starts)))
for w in (which, which + 1):
    if w + 1 >= len(starts): break
    a, b = starts[w], starts[w + 1]
    t0 = rows[a][1]
    last_end = {}
    busy = 0; ev = []
    print("merge %d: %.1f us from its first kernel to the next merge's first kernel" % (w, (rows[b][1] - t0) / 1e3))
    for n, s, e, qq in rows[a:b]:
        gap = (s - last_end[qq]) / 1e3 if qq in last_end else 0.0
        last_end[qq] = e
        ev += [(s, 1), (e, -1)]
        short = n.split("(")[0].replace("void ", "")[:60]
        print("  q%-3s %9.1f +%8.1f  gap %6.1f  %s" % (qq, (s - t0) / 1e3, (e - s) / 1e3, gap, short))
    ev.sort(); depth = 0; last = None
    for t, d in ev:
        if depth > 0: busy += t - last
        depth += d; last = t
    print("  some kernel running: %.1f us of %.1f" % (busy / 1e3, (rows[b][1] - t0) / 1e3))
PY
rm -rf gpurun_out/prof/trace_tl
