#!/usr/bin/env python3
"""One merge round as a timeline: every kernel between two consecutive k_chain launches of a rocprofv3 --kernel-trace database,
with its start offset, duration and the gap before it.   python tools/timeline.py results.db [which_round]"""
import sqlite3
import sys
sys.path.insert(0, "tools")
from prof_summary import short

con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, start, end, stream_id from kernels order by start").fetchall() if "stream_id" in [d[1] for d in con.execute("pragma table_info(kernels)")] else [r + (0,) for r in con.execute("select name, start, end from kernels order by start")]
chains = [i for i, r in enumerate(rows) if "k_chain" in r[0]]
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(chains) - 3
a, b = chains[which], chains[which + 1]
# start from the first kernel of the round: walk back to the tile histogram (a third argument: from this k_chain to the next, as is)
raw = len(sys.argv) > 3
while not raw and a > 0 and "k_tile_hist" not in rows[a][0]:
    a -= 1
if raw: b += 1
t0 = rows[a][1]
prev_end = {}
tot_k = tot_gap = 0.0
for name, s, e, st in rows[a:b]:
    if not raw and (("k_tile_hist" in name and s != t0) or "k_s_flag" in name):
        break
    gap = (s - prev_end[st]) / 1e3 if st in prev_end else 0.0
    prev_end[st] = e
    print("%9.1f us  +%7.1f us gap  %8.1f us  stream %s  %s" % ((s - t0) / 1e3, gap, (e - s) / 1e3, st, short(name)[:70]))
    tot_k += (e - s) / 1e3
    tot_gap += max(gap, 0.0)
print("kernels %.1f us, gaps %.1f us" % (tot_k, tot_gap))
