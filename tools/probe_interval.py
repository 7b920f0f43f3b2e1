"""Interval-sharded merge (ropebwt3_amd.multi.merge_interval) with W ranks as threads on ONE GPU: time per merge, rounds,
time inside the step kernels -- the protocol overhead without a real interconnect.   python tools/probe_interval.py W [n_reads] [index_symbols]"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import util
from ropebwt3_amd import Rb3Gpu, host, multi
W = int(sys.argv[1]) if len(sys.argv) > 1 else 2
NR = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
NI = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 26
rng = np.random.default_rng(5)
g = util.random_genome(rng, NI // 2 - 1)
h0 = Rb3Gpu(verbose=1)
t1 = util.make_text([g])
d, dtw = h0.sort_text(t1)
b1 = h0.dev_download(d, t1.size)
h0.dev_free(d); h0.dev_free(dtw)
st = rng.integers(0, len(g) - 150, size=NR)
r = np.stack([g[s:s + 150] for s in st])
m = rng.random(r.shape) < 0.01
r[m] = rng.integers(1, 5, size=int(m.sum()), dtype=np.uint8)
t2 = util.make_text(list(r))
sent = np.flatnonzero(t2 == 0).astype(np.int64)
# single-GPU reference point: the ordinary merge of the same batch
h0.from_plain(b1)
d, dtw = h0.sort_text(t2)
h0.merge_text_dev(d, dtw, t2.size, 2 * NR, commit=False)
h0.stats_reset(); t = time.perf_counter()
for _ in range(3): h0.merge_text_dev(d, dtw, t2.size, 2 * NR, commit=False)
print("one handle, ordinary merge: %.2f ms per merge of %d symbols (rank %.2f, rebuild %.2f)" % ((time.perf_counter() - t) / 3 * 1e3, t2.size, h0.stats()["ms_rank"] / 3, h0.stats()["ms_build"] / 3))
h0.close()
shared = multi.ThreadComm.Shared(W)
bounds = multi.interval_bounds(b1.size, W)
res = [None] * W
def run(rank):
    h = Rb3Gpu(verbose=1)
    comm = multi.ThreadComm(shared, rank, h)
    h.from_plain(b1[bounds[rank]:bounds[rank + 1]])
    d, dtw = h.sort_text(t2)
    sx = {}
    multi.merge_interval(h, comm, bounds, d, dtw, t2.size, sent, commit=False, stats=sx)
    h.stats_reset(); shared.barrier.wait(); t = time.perf_counter()
    for _ in range(3): multi.merge_interval(h, comm, bounds, d, dtw, t2.size, sent, commit=False, stats=sx)
    dt = (time.perf_counter() - t) / 3
    s = h.stats()
    res[rank] = (dt, sx["rounds"], s["ms_rank"] / 3, s["ms_build"] / 3, sx["rows_per_rank"][rank])
    h.close()
th = [threading.Thread(target=run, args=(k,)) for k in range(W)]
[t.start() for t in th]; [t.join() for t in th]
for k, x in enumerate(res):
    x = (x[0] * 1e3,) + x[1:]
    print("rank %d of %d (threads on one GPU): %.2f ms per merge, %d rounds, step kernels %.2f ms, rebuild %.2f ms, rows landing here %d" % ((k, W) + x))
