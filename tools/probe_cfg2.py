"""Scratch probe: config-2 style merge (two similar genomes), timing per split setting."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import util
from ropebwt3_amd import Rb3Gpu

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4400000
rate = float(sys.argv[2]) if len(sys.argv) > 2 else 0.001
ref = util.Reference()
rng = np.random.default_rng(1)
g0 = util.random_genome(rng, n)
g1 = util.mutate(np.random.default_rng(2), g0, rate)
t = time.time(); b1 = ref.bwt(util.make_text([g0]), 8); b2 = ref.bwt(util.make_text([g1]), 8); print("sais %.2fs" % (time.time() - t))
orc = util.Oracle()
t = time.time(); want = orc.merge(b1, b2, 8); print("oracle merge %.2fs" % (time.time() - t))
t = time.time(); rb, _ = orc.mg_rank(b1, b2, 8); print("oracle rank %.2fs" % (time.time() - t))
for sl in [-1, 8, 10, 14]:
    if sl == -1 and n > 1000000: continue
    h = Rb3Gpu(split_log2=sl, verbose=1)
    h.from_plain(b1)
    d = h.dev_upload(b2)
    h.merge_plain_dev(d, b2.size, commit=False)
    h.stats_reset()
    t = time.time()
    for _ in range(3): h.merge_plain_dev(d, b2.size, commit=False)
    dt = (time.time() - t) / 3
    st = h.stats()
    print("split_log2=%d: %.2f ms/merge (lf %.2f rank %.2f build %.2f) steps/merge=%d  -> %.3f Gsym/s" % (sl, dt*1e3, st['ms_lf']/3, st['ms_rank']/3, st['ms_build']/3, st['n_lf_steps']//3, b2.size/dt/1e9))
    pos, _ = h.mg_rank_plain(b2)
    print("   pos equal:", np.array_equal(pos, rb >> 6), " unset:", int((pos < 0).sum()))
    h.merge_plain(b2); print("   merged equal:", np.array_equal(h.export_plain(), want))
    h.dev_free(d); h.close()
