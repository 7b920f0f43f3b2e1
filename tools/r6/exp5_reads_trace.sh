#!/bin/bash
# GPU box, round 6: which kernel of the rebuild takes 45 instead of 10 ms per merge once the reads index has passed 2^32 symbols?  rocprofv3 kernel trace of
# `ropebwt3-amd build -L -d -m7g` on N reads; per-dispatch durations of the rebuild's kernels in launch order
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out/prof
N=${1:-30000000}; F=/dev/shm/rb3_reads_$N.txt
python tools/gen_reads.py $N $F > /dev/null
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/prof/trace_reads -o reads -- $R/ropebwt3_amd/ropebwt3-amd build -L -d -m7g -o /dev/shm/rb3_tr.fmd $F > $R/gpurun_out/prof/trace_reads.log 2>&1
cd $R
DB=$(ls gpurun_out/prof/trace_reads/*_results.db gpurun_out/prof/trace_reads/*/*_results.db 2>/dev/null | head -1)
python - "$DB" <<'PY' | tee gpurun_out/prof/r6_reads_rebuild_per_dispatch.txt
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = db.execute("select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, ks)).fetchall()
names = ("k_plane_group", "k_scan_place", "k_win_rows", "k_chain", "k_pos_finalize", "k_tile_hist", "k_lf2", "k_fill_regions")
per = {}
for n, a, b in rows:
    for k in names:
        if k in n:
            per.setdefault(k, []).append((b - a) / 1e3)
for k, v in per.items():
    print("%-18s %3d dispatches, us each: %s" % (k, len(v), " ".join("%.0f" % x for x in v[:40])))
PY
rm -rf gpurun_out/prof/trace_reads $F /dev/shm/rb3_tr.fmd
