"""Scratch probe: row-word walkers vs text-order walkers on a config-2 style merge (timing + parity)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import util
from ropebwt3_amd import Rb3Gpu, host

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4400000
rate = float(sys.argv[2]) if len(sys.argv) > 2 else 0.001
ngen = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rng = np.random.default_rng(1)
g0 = util.random_genome(rng, n)
idx = [g0] + [util.mutate(np.random.default_rng(100 + i), g0, rate) for i in range(ngen - 1)]
g1 = util.mutate(np.random.default_rng(2), g0, rate)
t = time.time(); b1 = host.build_bwt(util.make_text(idx)); t2 = util.make_text([g1]); print("sais %.2fs" % (time.time() - t))
b2 = host.build_bwt(t2)
h = Rb3Gpu(verbose=1); h.from_plain(b1)
d = h.dev_upload(b2)
d_bwt, d_tw = h.sort_text(t2)
assert np.array_equal(h.dev_download(d_bwt, b2.size), b2), "GPU sorter BWT differs"
def run(name, fn, reps=10):
    fn(); fn(); h.stats_reset()
    t = time.time()
    for _ in range(reps): fn()
    dt = (time.time() - t) / reps
    st = h.stats()
    print("%-28s %.3f ms/merge (lf %.3f chain %.3f rank %.3f build %.3f) steps=%d fb=%d -> %.3f Gsym/s" % (name, dt*1e3, st['ms_lf']/reps, st['ms_chain']/reps, st['ms_rank']/reps, st['ms_build']/reps, st['n_lf_steps']//reps, st['n_fallbacks'], b2.size/dt/1e9))
ref = None
for step in [192, 256, 384, 512]:
    _, w = host.build_bwt_walkers(t2, step)
    wt = host.walkers_text(t2, step)
    assert w.shape == wt.shape
    run("rows step %d (%d w)" % (step, w.shape[0]), lambda: h.merge_plain_dev_walkers(d, b2.size, w, commit=False))
    run("text step %d (%d w)" % (step, wt.shape[0]), lambda: h.merge_text_dev(d_bwt, d_tw, b2.size, wt, commit=False))
    p1, a1 = h.mg_rank_plain_walkers(b2, w)
    p2, a2 = h.mg_rank_text_dev(d_bwt, d_tw, b2.size, wt)
    print("   pos equal:", np.array_equal(p1, p2), "acc equal:", np.array_equal(a1, a2))
if os.environ.get("ORACLE"):
    orc = util.Oracle()
    rb, _ = orc.mg_rank(b1, b2, 8)
    print("pos == oracle:", np.array_equal(p2, rb >> 6))
h.merge_text_dev(d_bwt, d_tw, b2.size, wt, commit=True)
out_text = h.export_plain()
h2 = Rb3Gpu(verbose=1); h2.from_plain(b1); h2.merge_plain_walkers(b2, w)
print("merged index equal:", np.array_equal(out_text, h2.export_plain()))
