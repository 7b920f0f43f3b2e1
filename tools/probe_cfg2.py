"""Scratch probe: config-2 style merge (two similar genomes), timing per split setting."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import util
from ropebwt3_amd import Rb3Gpu, host

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4400000
rate = float(sys.argv[2]) if len(sys.argv) > 2 else 0.001
g0 = util.random_genome(np.random.default_rng(1), n)
g1 = util.mutate(np.random.default_rng(2), g0, rate)
t = time.time(); b1 = host.build_bwt(util.make_text([g0])); t2 = util.make_text([g1]); print("sais %.2fs" % (time.time() - t))
orc = util.Oracle()
b2 = host.build_bwt(t2)
rb, _ = orc.mg_rank(b1, b2, 8)
def run(name, h, fn, reps=5):
    fn(); h.stats_reset()
    t = time.time()
    for _ in range(reps): fn()
    dt = (time.time() - t) / reps
    st = h.stats()
    print("%-22s %.2f ms/merge (lf %.2f chain %.2f build %.2f) steps=%d fb=%d -> %.3f Gsym/s" % (name, dt*1e3, st['ms_lf']/reps, st['ms_chain']/reps, st['ms_build']/reps, st['n_lf_steps']//reps, st['n_fallbacks'], b2.size/dt/1e9))
for sl in [6, 7, 8, 9]:
    h = Rb3Gpu(split_log2=sl, verbose=1); h.from_plain(b1); d = h.dev_upload(b2)
    run("sa-order 2^%d" % sl, h, lambda: h.merge_plain_dev(d, b2.size, commit=False))
    h.dev_free(d); h.close()
h = Rb3Gpu(verbose=1); h.from_plain(b1); d = h.dev_upload(b2)
for step in [192, 256, 384, 512, 768, 1024]:
    _, w = host.build_bwt_walkers(t2, step)
    run("text step %d (%d w)" % (step, w.shape[0]), h, lambda: h.merge_plain_dev_walkers(d, b2.size, w, commit=False))
_, w = host.build_bwt_walkers(t2, 512)
h.mg_begin(d, b2.size); h.mg_walk(w); p, ln = h.mg_pos_ptr()
pos = np.empty(ln, dtype=np.int64); h._chk(h._lib.rb3gpu_dev_download(h._h, pos.ctypes.data, p, ln * 8), "dl"); h.mg_finish(False)
print("pos equal:", np.array_equal(pos, rb >> 6))
