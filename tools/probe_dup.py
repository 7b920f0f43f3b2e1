"""Merge an exact duplicate of the indexed genome (and of one of K relatives): the pathological case for the settle phase.
  python tools/probe_dup.py [K]"""
import sys, time, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tests import util
from ropebwt3_amd import Rb3Gpu, host
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1
g0 = util.random_genome(np.random.default_rng(1), 4400000)
h = Rb3Gpu(verbose=1)
rel = [g0] + [util.mutate(np.random.default_rng(100 + k), g0, 0.001) for k in range(K - 1)]
for k, g in enumerate(rel):
    t = util.make_text([g])
    d, dtw = h.sort_text(t)
    if k == 0: h.from_plain_dev(d, t.size)
    else: h.merge_text_dev(d, dtw, t.size, host.walkers_text(t, 384), commit=True)
    h.dev_free(d); h.dev_free(dtw)
for name, g in (("0.1 % divergent genome", util.mutate(np.random.default_rng(999), g0, 0.001)), ("exact duplicate of the first genome", g0), ("exact duplicate of the last genome", rel[-1])):
    t = util.make_text([g]); w = host.walkers_text(t, 384)
    d, dtw = h.sort_text(t)
    b = h.dev_download(d, t.size)
    d2 = h.dev_upload(b)
    for entry, fn in (("text", lambda: h.merge_text_dev(d, dtw, t.size, w, commit=False)), ("plain", lambda: h.merge_plain_dev(d2, t.size, commit=False))):
        fn(); h.stats_reset(); t0 = time.perf_counter()
        for _ in range(3): fn()
        dt = (time.perf_counter() - t0) / 3; st = h.stats()
        print("K=%d %-38s %-5s %.3f ms per merge (rank %.3f, chain %.3f, rebuild %.3f) steps/row %.2f fallbacks %d" % (K, name, entry, dt * 1e3, st["ms_rank"] / 3, st["ms_chain"] / 3, st["ms_build"] / 3, st["n_lf_steps"] / 3 / t.size, st["n_fallbacks"]))
    h.dev_free(d); h.dev_free(dtw); h.dev_free(d2)
