#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (sqlite) outputs into small text/JSON files for profiles/.

    python tools/prof_summary.py stats  <results.db> <out.txt>          # --kernel-trace --stats
    python tools/prof_summary.py pmc    <results.db> <COUNTER> <out.txt> # one --pmc pass
"""
import json
import sqlite3
import sys


def short(name):
    if "k_chain" in name:
        import re
        m = re.search(r"k_chain<([^>]*)>", name)
        if m:
            f = [x.strip() in ("true", "(bool)1") for x in m.group(1).split(",")]
            return "k_chain<%s,%s,%s>" % ("list" if f[0] else "auto", "dense" if f[1] else "mixed", "tent" if len(f) > 2 and f[2] else "plain")
        return "k_chain"
    if "k_reb_group" in name:
        import re
        m = re.search(r"k_reb_group<(\d+), *(\d+)", name) or re.search(r"k_reb_groupILi(\d+)ELi(\d+)", name)
        return "k_reb_group<%s,%s>" % (m.group(1), m.group(2)) if m else "k_reb_group"
    if "k_pos_finalize_check_rowsN" in name:
        return "k_pos_finalize_check_rowsN"
    if "k_pos_finalize_check_rows2" in name:
        return "k_pos_finalize_check_rows2"
    if "k_pos_finalize_check_rows" in name:
        return "k_pos_finalize_check_rows"
    for k in ("k_plane_group", "k_place_pg", "k_part_scatter", "k_part_place"):
        if k in name:
            return k + ("<listed>" if "<true>" in name or "ILb1" in name else "")
    if "k_export_runs" in name:
        return "k_export_runs"
    for k in ("k_events", "k_ssa_walk", "k_ssa_link", "k_ssa_final"):
        if k in name:
            return k
    for k in ("k_pos_finalize_check", "k_pass1w", "k_pass2w", "k_win_rows", "k_decide"):
        if k in name:
            return k + ("<plain>" if "<true>" in name else "<merge>" if "<false>" in name else "")
    for k in ("k_resolve", "k_pos_finalize", "k_pass1", "k_pass2", "k_lf2", "k_tile_hist", "k_scan_chunk_totals", "k_scan_chunks", "k_scan_records",
              "k_group_rows", "k_export_plain", "k_rank_batch", "k_pos_check", "k_jump", "k_ckpt"):
        if k in name:
            return k + ("<plain>" if ("ILb1" in name or "<true>" in name) else "<merge>" if ("ILb0" in name or "<false>" in name) else "")
    return name[:60]


def stats(db, out):
    con = sqlite3.connect(db)
    rows = con.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    with open(out, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats summary (durations in us)\n")
        f.write("%-28s %8s %14s %12s %12s %12s %7s\n" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
        for name, n, s, a, mn, mx in rows:
            f.write("%-28s %8d %14.1f %12.2f %12.2f %12.2f %6.2f%%\n" % (short(name), n, s / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))
    print(open(out).read())


def pmc(db, counter, out):
    con = sqlite3.connect(db)
    cols = [d[0] for d in con.execute("select * from counters_collection limit 1").description]
    ncol = "counter_name" if "counter_name" in cols else "name"
    kcol = "kernel_name" if "kernel_name" in cols else "name"
    q = "select %s, count(*), sum(value), avg(value) from counters_collection where %s = ? group by %s order by sum(value) desc" % (kcol, ncol, kcol)
    rows = con.execute(q, (counter,)).fetchall()
    with open(out, "w") as f:
        f.write("# rocprofv3 --pmc %s summary (raw counter units; see MI355X_MICROARCH.md HBM section)\n" % counter)
        f.write("%-28s %8s %18s %18s\n" % ("kernel", "calls", "sum", "avg_per_dispatch"))
        for name, n, s, a in rows:
            f.write("%-28s %8d %18.1f %18.1f\n" % (short(name), n, s, a))
    print(open(out).read())
    return rows


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "pmc":
        pmc(sys.argv[2], sys.argv[3], sys.argv[4])
    elif sys.argv[1] == "traffic":  # traffic <fetch.db> <write.db> <kernel-substring> <out.json>
        import sqlite3 as sq
        def avg(db, counter):
            con = sq.connect(db)
            r = con.execute("select avg(value), count(*) from counters_collection where counter_name = ? and kernel_name like ?", (counter, "%" + sys.argv[4] + "%")).fetchone()
            return r
        f, nf = avg(sys.argv[2], "FETCH_SIZE")
        w, nw = avg(sys.argv[3], "WRITE_SIZE")
        d = {"kernel": sys.argv[4], "dispatches": nf, "FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w,
             "hbm_bytes_per_launch": int((2 * f + w) * 1024),
             "correction": "MI355X_MICROARCH.md HBM section: gfx950 FETCH_SIZE tallies 128-B requests at 64 B -> x2 (upper bound: smaller loads are not halved); WRITE_SIZE as reported. Calibrated earlier this round with k_lf2 in the same kind of passes: 8.8 MB streamed in -> FETCH_SIZE 4.5 MB, 70.4 MB streamed out -> WRITE_SIZE 70.4 MB."}
        json.dump(d, open(sys.argv[5], "w"), indent=1)
        print(d)
    elif sys.argv[1] == "cols":
        con = sqlite3.connect(sys.argv[2])
        print([d[0] for d in con.execute("select * from counters_collection limit 1").description])
        print(con.execute("select * from counters_collection limit 3").fetchall())
