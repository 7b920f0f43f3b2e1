"""kernel experiment: k_chain on the large-index leg with the record stores removed / redirected (variant libraries built with
-DRB3_EXP_NOREC=1|2; their merges fail validation, only the kernel time counts):   python tools/probe_norec.py lib.so ..."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import util
from ropebwt3_amd import Rb3Gpu
n_index, n_reads = 1 << 30, 1000000
rng = np.random.default_rng(21)
g = util.random_genome(rng, n_index // 2 - 1)
st = rng.integers(0, len(g) - 150, size=n_reads)
r = np.stack([g[s:s + 150] for s in st])
t2 = util.make_text(list(r))
for lib in [None] + sys.argv[1:]:
    h = Rb3Gpu(verbose=0, lib=lib)
    d, d_tw = h.sort_text(util.make_text([g]))
    h.dev_free(d_tw)
    h.from_plain_dev(d, n_index)
    h.dev_free(d)
    d, d_tw = h.sort_text(t2)
    for rep in range(3):
        if rep == 1:
            h.stats_reset()
        try:
            h.merge_text_dev(d, d_tw, t2.size, 2 * n_reads, commit=False)
        except Exception as e:
            pass
    s = h.stats()
    print("%-14s k_chain %.3f ms per launch (%d launches), %.2f M steps per launch" % ((lib or "release").split("/")[-1], s["ms_chain"] / max(1, s["n_rank_launches"]), s["n_rank_launches"], s["n_lf_steps"] / max(1, s["n_rank_launches"]) / 1e6), flush=True)
    h.close()
