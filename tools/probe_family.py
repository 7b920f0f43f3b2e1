"""Scratch probe: merge genome K+1 into an index of K close relatives (star phylogeny), per-merge
phase times and LF step counts.   python tools/probe_family.py K L"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import util
from ropebwt3_amd import Rb3Gpu, host

K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
L = int(sys.argv[2]) if len(sys.argv) > 2 else 4400000
step = int(sys.argv[3]) if len(sys.argv) > 3 else 384
g0 = util.random_genome(np.random.default_rng(1), L)
h = Rb3Gpu(verbose=1)
t0 = time.time()
for k in range(K + 1):
    g = util.mutate(np.random.default_rng(100 + k), g0, 0.001)
    b, w = host.build_bwt_walkers(util.make_text([g]), step)
    if k == 0:
        h.from_plain(b)
        continue
    h.stats_reset()
    t = time.time()
    h.merge_plain_walkers(b, w)
    dt = time.time() - t
    st = h.stats()
    if k < 4 or k % 8 == 0 or k == K:
        print("k=%3d  %.2f ms (h2d %.2f lf %.2f rank %.2f chain %.2f build %.2f) steps=%d (%.2f/row) fb=%d index %.1f MB" % (k, dt * 1e3, st["ms_h2d"], st["ms_lf"], st["ms_rank"],
              st["ms_chain"], st["ms_build"], st["n_lf_steps"], st["n_lf_steps"] / b.size, st["n_fallbacks"], st["bytes_index"] / 1e6), flush=True)
print("total %.1f s" % (time.time() - t0))
