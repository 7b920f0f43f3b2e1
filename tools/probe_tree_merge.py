"""the tree-merge step of the partitioned build on one GPU: the index of K genomes of the synthetic mtb star merged into the index of
K others (rb3gpu_export_plain_dev + rb3gpu_merge_plain_dev, what bench.py --gpus N does between ranks), device time per phase:
    python tools/probe_tree_merge.py [K]"""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import gen_mtb
from ropebwt3_amd import Rb3Gpu, Sorter, host, walker_step
K = int(sys.argv[1]) if len(sys.argv) > 1 else 12
files = gen_mtb.generate(2 * K, 4400000, "/tmp/mtb_star_4400000")
hs = [Rb3Gpu(verbose=1), Rb3Gpu(verbose=1)]
srt = Sorter(0)
for i, fn in enumerate(files):
    h = hs[i // K]
    (n_seq, t), = list(host.read_batches(fn, False, 1 << 40))
    srt.upload(t); d, dtw = srt.sort_uploaded(t.size)
    if i % K == 0: h.from_plain_dev(d, t.size)
    else: h.merge_text_dev(d, dtw, t.size, host.walkers_text(t, walker_step(0, t.size, n_seq)), commit=True)
    srt.release(d)
a, b = hs
n2 = b.get_tot()
p = ctypes.c_void_p(); a._chk(a._lib.rb3gpu_dev_alloc(a._h, n2, ctypes.byref(p)), "alloc")
b.export_plain_dev(p.value)
for rep in range(3):
    a.stats_reset()
    t0 = time.time()
    a.merge_plain_dev(p.value, n2, commit=False)
    dt = time.time() - t0
    s = a.stats()
    print("merge of %d symbols into %d: %.1f ms wall; lf %.2f rank %.2f (k_chain %.2f) rebuild %.2f ms; fallbacks %d; %.2f Gbp/s" % (n2, a.get_tot(), dt * 1e3, s["ms_lf"], s["ms_rank"], s["ms_chain"], s["ms_build"], s["n_fallbacks"], n2 / dt / 1e9), flush=True)
t0 = time.time(); a.merge_plain_dev(p.value, n2, commit=True); print("committed: %.1f ms wall, rebuild %.2f ms, index %d symbols" % ((time.time() - t0) * 1e3, a.stats()["ms_build"], a.get_tot()))
t0 = time.time(); a.merge_index(b); print("rb3gpu_merge_index (handle to handle): %.1f ms wall" % ((time.time() - t0) * 1e3))
