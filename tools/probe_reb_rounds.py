"""per-round cost of the rebuild in a K-genome build of the synthetic mtb star: milliseconds, groups through the run-space kernels
and groups handed on to the window kernels, round by round:  python tools/probe_reb_rounds.py [K]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import gen_mtb
from ropebwt3_amd import Rb3Gpu, Sorter, host, walker_step
K = int(sys.argv[1]) if len(sys.argv) > 1 else 152
files = gen_mtb.generate(K, 4400000, "/tmp/mtb_star_4400000")
h = Rb3Gpu(verbose=0)
srt = Sorter(0)
prev = None
tot = 0.0
for rep in range(2):
    for i, fn in enumerate(files):
        (n_seq, t), = list(host.read_batches(fn, False, 1 << 40))
        srt.upload(t); d, dtw = srt.sort_uploaded(t.size)
        if i == 0: h.from_plain_dev(d, t.size)
        else: h.merge_text_dev(d, dtw, t.size, host.walkers_text(t, walker_step(0, t.size, n_seq)), commit=True)
        srt.release(d)
        s = h.stats()
        if prev is not None and i > 0 and rep == 1:
            db = s["ms_build"] - prev["ms_build"]
            tot += db
            if i < 24 or i % 8 == 0:
                print("round %3d  rebuild %.3f ms  rank %.3f  chain %.3f  groups: run space %d, window kernels %d, index %.1f MB" % (i, db, s["ms_rank"] - prev["ms_rank"], s["ms_chain"] - prev["ms_chain"],
                      s["n_reb_groups"] - prev["n_reb_groups"], s["n_reb_groups_window"] - prev["n_reb_groups_window"], s["bytes_index"] / 1e6), flush=True)
        prev = s
print("rebuild total %.1f ms" % tot)
