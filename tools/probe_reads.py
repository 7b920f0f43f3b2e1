"""Scratch probe: many-short-strings regime (reads), throughput of the chain kernel."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import util
from ropebwt3_amd import Rb3Gpu, host
G = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
NR = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
rng = np.random.default_rng(1)
g = util.random_genome(rng, G)
st = rng.integers(0, G - 150, size=2 * NR)
def reads(idx):
    r = np.stack([g[s:s + 150] for s in idx])
    m = rng.random(r.shape) < 0.01
    r[m] = rng.integers(1, 5, size=int(m.sum()), dtype=np.uint8)
    return list(r)
t = time.time()
b1 = host.build_bwt(util.make_text(reads(st[:NR]))); b2 = host.build_bwt(util.make_text(reads(st[NR:])))
print("sais %.1fs; B1 %d B2 %d symbols, %d strings" % (time.time() - t, b1.size, b2.size, (b2 == 0).sum()))
h = Rb3Gpu(verbose=1); h.from_plain(b1); d = h.dev_upload(b2)
st0 = h.stats(); print("index bytes %.1f MB (%.3f B/sym)" % (st0['bytes_index'] / 1e6, st0['bytes_index'] / b1.size))
h.merge_plain_dev(d, b2.size, commit=False); h.stats_reset()
R = 5
t = time.time()
for _ in range(R): h.merge_plain_dev(d, b2.size, commit=False)
dt = (time.time() - t) / R
s = h.stats()
print("merge %.2f ms: lf %.2f chain %.2f build %.2f | %.2f Gsym/s; chain: %.2f G steps/s, algorithmic %.0f GB/s (208 B/step)" % (
    dt * 1e3, s['ms_lf'] / R, s['ms_chain'] / R, s['ms_build'] / R, b2.size / dt / 1e9, b2.size / (s['ms_chain'] / R * 1e-3) / 1e9, 208 * b2.size / (s['ms_chain'] / R * 1e-3) / 1e9))
orc = util.Oracle()
if b2.size < 80e6:
    rb, _ = orc.mg_rank(b1, b2, 32)
    pos, _ = h.mg_rank_plain(b2)
    print("pos equal:", np.array_equal(pos, rb >> 6))
