"""Scratch probe for kernel ablations: build the index of K genomes of the synthetic family with the normal library, then
time the merge of one more genome (result discarded) with every variant build of librb3gpu.so named on the command line
(tools/build_variant.sh; errors from deliberately wrong kernels are ignored).
  python tools/probe_rebuild.py K [variant.so ...] [key=value ...]      (key=value: rb3gpu_tune switches for every handle)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import util
from ropebwt3_amd import Rb3Gpu, host
K = int(sys.argv[1]); L = 4400000
libs = [None] + [a for a in sys.argv[2:] if "=" not in a]
tunes = [a.split("=") for a in sys.argv[2:] if "=" in a and not a.startswith("W=")]
Ws = [int(a[2:]) for a in sys.argv[2:] if a.startswith("W=")] or [384]
g0 = util.random_genome(np.random.default_rng(1), L)
h = Rb3Gpu(verbose=1)
for k in range(K):
    t = util.make_text([util.mutate(np.random.default_rng(100 + k), g0, 0.001)])
    d, dtw = h.sort_text(t)
    if k == 0: h.from_plain_dev(d, t.size)
    else: h.merge_text_dev(d, dtw, t.size, host.walkers_text(t, 384), commit=True)
    h.dev_free(d); h.dev_free(dtw)
n = h.get_tot()
d_plain = h._dev_alloc(n) if hasattr(h, "_dev_alloc") else None
if d_plain is None:
    import ctypes
    p = ctypes.c_void_p()
    h._chk(h._lib.rb3gpu_dev_alloc(h._h, n, ctypes.byref(p)), "alloc"); d_plain = p.value
h.export_plain_dev(d_plain)
t = util.make_text([util.mutate(np.random.default_rng(999), g0, 0.001)])
d, dtw = h.sort_text(t)
print("index of %d genomes: %d symbols, %.1f MB" % (K, n, h.stats()["bytes_index"] / 1e6))
for lib, W in [(l, W) for l in libs for W in Ws]:
    w = host.walkers_text(t, W)
    h2 = Rb3Gpu(verbose=0, lib=lib)
    for k, v in tunes: h2.tune(k, int(v))
    h2.from_plain_dev(d_plain, n)
    for rep in range(6):
        if rep == 1: h2.stats_reset()
        try: h2.merge_text_dev(d, dtw, t.size, w, commit=False)
        except Exception as e: pass
    st = h2.stats()
    print("%-12s W=%d: rebuild %.3f ms, rank %.3f ms (chain %.3f, %.2f M steps) per merge; groups to the window kernels %d of %d" % ((lib or "default").split("/")[-1], W, st["ms_build"] / 5, st["ms_rank"] / 5, st["ms_chain"] / 5,
          st["n_lf_steps"] / 5e6, st["n_reb_groups_window"] // 5, st["n_reb_groups"] // 5))
    h2.close()
