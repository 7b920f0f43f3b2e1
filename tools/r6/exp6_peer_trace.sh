#!/bin/bash
# GPU box, round 6: the lock-step rounds of the interval-sharded merge at world W on this device, as peer rounds (mode 0) and driven by the host (mode 1):
# rocprofv3 kernel trace of tools/probe_sh_peer.py; per mode the k_sh_round dispatches -- how long they take, how much of the time some round kernel runs
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out/prof
N=${1:-1000000}; W=${2:-2}
export TMPDIR=/tmp
for MODE in 0 1; do
  cd /tmp
  RB3_PROBE_WORLDS=$W RB3_PROBE_MODES=$MODE rocprofv3 --kernel-trace -d $R/gpurun_out/prof/trace_peer$MODE -o peer -- python $R/tools/probe_sh_peer.py $N > $R/gpurun_out/prof/trace_peer$MODE.log 2>&1
  cd $R
  tail -1 gpurun_out/prof/trace_peer$MODE.log
  DB=$(ls gpurun_out/prof/trace_peer$MODE/*_results.db gpurun_out/prof/trace_peer$MODE/*/*_results.db 2>/dev/null | head -1)
  python - "$DB" $MODE <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in db.execute("pragma table_info(%s)" % kd)]
qcol = "queue_id" if "queue_id" in cols else None
rows = db.execute("select s.kernel_name, d.start, d.end%s from %s d join %s s on d.kernel_id = s.id order by d.start" % (", d." + qcol if qcol else ", 0", kd, ks)).fetchall()
sh = [(a, b, q) for n, a, b, q in rows if "k_sh_round" in n]
if not sh:
    print("no k_sh_round dispatches"); sys.exit(0)
# the timed merges are the last 3/4 of the dispatches (one warm-up merge + three timed)
sh = sh[len(sh) // 4:]
dur = [(b - a) / 1e3 for a, b, q in sh]
span = (max(b for a, b, q in sh) - min(a for a, b, q in sh)) / 1e3
# time during which at least one round kernel runs
ev = sorted([(a, 1) for a, b, q in sh] + [(b, -1) for a, b, q in sh])
busy, depth, last = 0, 0, None
for t, d in ev:
    if depth > 0: busy += t - last
    depth += d; last = t
queues = sorted(set(q for a, b, q in sh))
print("mode %s: %d k_sh_round dispatches on %d queues, %.1f us each (min %.1f, max %.1f); span %.0f us, some round kernel running %.0f us (%.0f %%)" % (
    sys.argv[2], len(sh), len(queues), sum(dur) / len(dur), min(dur), max(dur), span, busy / 1e3, 100.0 * busy / 1e3 / span))
for q in queues[:2]:
    mine = [(a, b) for a, b, qq in sh if qq == q][:12]
    print("  queue %s, first dispatches (start offset us, duration us): %s" % (q, " ".join("%.0f+%.0f" % ((a - mine[0][0]) / 1e3, (b - a) / 1e3) for a, b in mine)))
PY
  rm -rf gpurun_out/prof/trace_peer$MODE
done
