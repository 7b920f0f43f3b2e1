"""What the device-made walker list of the BWT-only entry point looks like (k_b2_pick rule restated in numpy on the true
suffix array), and k_chain's time with it:   python tools/probe_b2list.py [S] [W] [FIRST]"""
import sys, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tests import util
from ropebwt3_amd import Rb3Gpu, host
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
W = int(sys.argv[2]) if len(sys.argv) > 2 else 320
FIRST = int(sys.argv[3]) if len(sys.argv) > 3 else 192
g0 = util.random_genome(np.random.default_rng(1), 4400000); g1 = util.mutate(np.random.default_rng(2), g0, 0.001)
b1 = host.build_bwt(util.make_text([g0])); t2 = util.make_text([g1])
h = Rb3Gpu(verbose=1); h.from_plain(b1)
d, isa = h.bwt_from_text(t2, 1)
n = t2.size
sa = np.empty(n, dtype=np.int64); sa[isa] = np.arange(n)
ends = np.flatnonzero(t2 == 0); m2 = len(ends)
rows = np.arange(m2, n, 1 << S)
tp = sa[rows]                                   # text position of every splitter's suffix
# D = distance from the start of its string; G = position in the concatenation of the strings (here: text order = string order)
starts = np.concatenate([[0], ends[:-1] + 1])
sid = np.searchsorted(ends, tp)
D = tp - starts[sid]
ln = ends - starts + 1
ok = (D > 0) & (D + 1 + FIRST <= ln[sid])
G = tp
off = G % W
cand = ok & (off < FIRST)
best = {}
order = np.argsort(off[cand], kind="stable")
bk = (G[cand] // W)[order]
first = np.unique(bk, return_index=True)[1]
pick_pos = np.sort(G[cand][order][first])
gaps = np.diff(pick_pos)
print("S=%d W=%d FIRST=%d: %d splitters, %d windows, %d picks, gaps min %d mean %.0f p99 %d max %d; empty windows %d" % (S, W, FIRST, len(rows), n // W + 1, len(pick_pos), gaps.min(), gaps.mean(), np.percentile(gaps, 99), gaps.max(), n // W + 1 - len(pick_pos)))
# the same list through the entry point that takes a host list, and the device-made one
w = []
pp = np.sort(G[cand][order][first])
for j, e in enumerate(ends):
    mine = pp[(pp >= starts[j]) & (pp < e)]
    prev = -1
    for p in mine:
        w.append((isa[p], -1, (1 << 62) if prev < 0 else p - prev, 0)); prev = p
    w.append((j, -2, (1 << 62) if prev < 0 else e - prev, 0))
w = np.array(w, dtype=np.int64)
for name, fn in (("host copy of the device rule", lambda: h.merge_plain_dev_walkers(d, n, w, commit=False)), ("device-made list", lambda: h.merge_plain_dev(d, n, commit=False)),
                 ("host copy again", lambda: h.merge_plain_dev_walkers(d, n, w, commit=False))):
    for i in range(3): fn()
    h.stats_reset()
    for i in range(10): fn()
    st = h.stats()
    print("%-30s k_chain %.3f ms, rank phase %.3f ms, steps %.2fM" % (name, st["ms_chain"] / 10, st["ms_rank"] / 10, st["n_lf_steps"] / 10 / 1e6))
for shift in (1, 5):
    w2 = []
    for j, e in enumerate(ends):
        mine = pp[(pp >= starts[j]) & (pp < e)] + shift
        prev = -1
        for p in mine:
            w2.append((isa[p], -1, (1 << 62) if prev < 0 else p - prev, 0)); prev = p
        w2.append((j, -2, (1 << 62) if prev < 0 else e - prev, 0))
    w2 = np.array(w2, dtype=np.int64)
    for i in range(3): h.merge_plain_dev_walkers(d, n, w2, commit=False)
    h.stats_reset()
    for i in range(10): h.merge_plain_dev_walkers(d, n, w2, commit=False)
    st = h.stats()
    print("same picks shifted by %d text positions: k_chain %.3f ms, steps %.2fM" % (shift, st["ms_chain"] / 10, st["n_lf_steps"] / 10 / 1e6))
