"""Scratch probe: reads regime, row-word auto mode vs text-order per-string walkers."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import util
from ropebwt3_amd import Rb3Gpu, host
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
rng = np.random.default_rng(11)
g = util.random_genome(rng, 10 * n_reads)
st = rng.integers(0, len(g) - 150, size=2 * n_reads)
def reads(idx):
    r = np.stack([g[s:s + 150] for s in idx])
    m = rng.random(r.shape) < 0.01
    r[m] = rng.integers(1, 5, size=int(m.sum()), dtype=np.uint8)
    return list(r)
b1 = host.build_bwt(util.make_text(reads(st[:n_reads])))
t2 = util.make_text(reads(st[n_reads:]))
b2 = host.build_bwt(t2.copy())
h = Rb3Gpu(verbose=1); h.from_plain(b1)
d = h.dev_upload(b2)
d_bwt, d_tw = h.sort_text(t2)
def run(name, fn, reps=5):
    fn(); fn(); h.stats_reset()
    t = time.time()
    for _ in range(reps): fn()
    dt = (time.time() - t) / reps
    s = h.stats()
    print("%-28s %.3f ms/merge (lf %.3f chain %.3f rank %.3f build %.3f) steps=%d -> %.3f Gsym/s" % (name, dt*1e3, s['ms_lf']/reps, s['ms_chain']/reps, s['ms_rank']/reps, s['ms_build']/reps, s['n_lf_steps']//reps, b2.size/dt/1e9), flush=True)
run("auto rows", lambda: h.merge_plain_dev(d, b2.size, commit=False))
run("text per-string", lambda: h.merge_text_dev(d_bwt, d_tw, b2.size, 2 * n_reads, commit=False))
run("rows per-string", lambda: h.merge_plain_dev_walkers(d, b2.size, 2 * n_reads, commit=False))
