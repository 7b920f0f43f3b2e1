"""Scratch: time k_chain alone through the staged API (no validation), for store/no-store experiments."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import util
from ropebwt3_amd import Rb3Gpu, host
g0 = util.random_genome(np.random.default_rng(1), 4400000); g1 = util.mutate(np.random.default_rng(2), g0, 0.001)
b1 = host.build_bwt(util.make_text([g0])); b2, w = host.build_bwt_walkers(util.make_text([g1]), 384)
h = Rb3Gpu(verbose=0); h.from_plain(b1); d = h.dev_upload(b2)
for rep in range(3):
    h.mg_begin(d, b2.size); h.stats_reset(); h.mg_walk(w); st = h.stats(); print("staged (no tentative) chain %.3f ms steps %d" % (st['ms_chain'], st['n_lf_steps']))
    try: h.mg_finish(False)
    except Exception as e: pass
rng = np.random.default_rng(1); G = 2000000; NR = 200000
g = util.random_genome(rng, G); st_ = rng.integers(0, G - 150, size=2 * NR)
rd = lambda idx: list(np.stack([g[s:s + 150] for s in idx]))
b1 = host.build_bwt(util.make_text(rd(st_[:NR]))); b2 = host.build_bwt(util.make_text(rd(st_[NR:])))
h2 = Rb3Gpu(verbose=0); h2.from_plain(b1); d2 = h2.dev_upload(b2)
for rep in range(3):
    h2.mg_begin(d2, b2.size); h2.stats_reset(); h2.mg_walk(None); st = h2.stats(); print("reads chain %.3f ms -> %.2f G steps/s" % (st['ms_chain'], b2.size / st['ms_chain'] / 1e6))
    try: h2.mg_finish(False)
    except Exception as e: pass
