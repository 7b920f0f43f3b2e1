"""k_chain's per-wave statistics (a -DRB3_PROF build of the library: RB3GPU_LIB=ropebwt3_amd/prof/prof.so) over the last rounds of a K-genome build"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import gen_mtb
from ropebwt3_amd import Rb3Gpu, Sorter, host, walker_step
K = int(sys.argv[1]) if len(sys.argv) > 1 else 152
files = gen_mtb.generate(K, 4400000, "/tmp/mtb_star_4400000")
h = Rb3Gpu(verbose=0)
srt = Sorter(0)
for i, fn in enumerate(files):
    (n_seq, t), = list(host.read_batches(fn, False, 1 << 40))
    srt.upload(t); d, dtw = srt.sort_uploaded(t.size)
    if i == 0: h.from_plain_dev(d, t.size)
    else: h.merge_text_dev(d, dtw, t.size, host.walkers_text(t, walker_step(0, t.size, n_seq)), commit=True)
    srt.release(d)
