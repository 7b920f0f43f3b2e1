"""Scratch probe for kernel ablations: time the rebuild of one merge into a K-genome index with whatever build of
librb3gpu.so RB3GPU_LIB names (results are discarded; errors from deliberately wrong kernels are ignored).
  python tools/probe_rebuild.py prep K   -> /tmp/probe_runs.npy, /tmp/probe_b2.npy, /tmp/probe_w.npy (normal library)
  python tools/probe_rebuild.py time"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import util
from ropebwt3_amd import Rb3Gpu, host
if sys.argv[1] == "prep":
    K = int(sys.argv[2]); L = 4400000
    g0 = util.random_genome(np.random.default_rng(1), L)
    h = Rb3Gpu(verbose=1)
    for k in range(K):
        b, w = host.build_bwt_walkers(util.make_text([util.mutate(np.random.default_rng(100 + k), g0, 0.001)]), 384)
        if k == 0: h.from_plain(b)
        else: h.merge_plain_walkers(b, w)
    runs = h.export_runs()
    np.save("/tmp/probe_runs.npy", np.array([(l << 3) | c for c, l in runs], dtype=np.uint64))
    b, w = host.build_bwt_walkers(util.make_text([util.mutate(np.random.default_rng(999), g0, 0.001)]), 384)
    np.save("/tmp/probe_b2.npy", b); np.save("/tmp/probe_w.npy", w)
else:
    arr = np.load("/tmp/probe_runs.npy"); b = np.load("/tmp/probe_b2.npy"); w = np.load("/tmp/probe_w.npy")
    h = Rb3Gpu(verbose=0)
    h._chk(h._lib.rb3gpu_from_runs(h._h, arr.size, arr.ctypes.data), "from_runs")
    d = h.dev_upload(b)
    for rep in range(6):
        if rep == 1: h.stats_reset()
        try: h.merge_plain_dev_walkers(d, b.size, w, commit=False)
        except Exception as e: pass
    st = h.stats()
    print("%s: rebuild %.3f ms, rank %.3f ms per merge" % (os.environ.get("RB3GPU_LIB", "default").split("/")[-1], st["ms_build"] / 5, st["ms_rank"] / 5))
