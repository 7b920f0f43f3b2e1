"""Scratch probe: per-step latency of one sequential LF chain (no splitting) vs index size."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import util
from ropebwt3_amd import Rb3Gpu, host
for n in [20000, 100000, 400000, 1600000]:
    g0 = util.random_genome(np.random.default_rng(1), n)
    g1 = util.mutate(np.random.default_rng(2), g0, 0.001)
    b1 = host.build_bwt(util.make_text([g0])); b2 = host.build_bwt(util.make_text([g1], rev=False))
    h = Rb3Gpu(split_log2=-1, verbose=1); h.from_plain(b1); d = h.dev_upload(b2)
    h.merge_plain_dev(d, b2.size, commit=False); h.stats_reset()
    h.merge_plain_dev(d, b2.size, commit=False)
    st = h.stats()
    print("n=%8d: chain %.3f ms for %d sequential steps -> %.1f ns/step (index %.2f MB)" % (n, st['ms_chain'], b2.size, st['ms_chain']*1e6/b2.size, st['bytes_index']/1e6))
    h.dev_free(d); h.close()
