"""Where does the interval-sharded build part from the plain one at scale?  N reads in three batches: plain build in one handle, sharded
build (Shard: split after the first batch, two sharded merges) in another; compared BEFORE and AFTER the gather."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ropebwt3_amd import Rb3Gpu, Shard, Sorter, host
from tests import util
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rng = np.random.default_rng(5)
g = util.random_genome(rng, 50000000)
def batch(n):
    st = rng.integers(0, len(g) - 150, size=n)
    idx = st[:, None] + np.arange(150)[None, :]
    r = g[idx]
    out = np.zeros((n, 151), dtype=np.uint8)
    out[:, :150] = r
    return out.reshape(-1)   # forward strands only: n strings of 150 + sentinel
bs = [batch(N // 3) for _ in range(3)]
a, b = Rb3Gpu(verbose=1), Rb3Gpu(verbose=1)
sh = None
import threading
stop = False
def hammer():   # what the CLI's sorter thread does beside the merges: suffix sorting of the next batch, on its own stream
    s2 = Sorter(0)
    while not stop:
        s2.upload(bs[0])
        d2, _ = s2.sort_uploaded(bs[0].size)
        s2.release(d2)
th = threading.Thread(target=hammer)
if os.environ.get("HAMMER"): th.start()
for i, t in enumerate(bs):
    d, dtw = a.sort_text(t)
    sent = np.flatnonzero(t == 0)
    if i == 0:
        a.from_plain_dev(d, t.size); b.from_plain_dev(d, t.size)
    else:
        a.merge_plain_dev(d, t.size)
        if sh is None: sh = Shard(b, [0] * W)
        sh.merge(d, dtw, t.size, sent)
        print("after batch %d: plain build %d symbols, bounds %s" % (i, a.get_tot(), sh.bounds().tolist()), flush=True)
    a.dev_free(d); a.dev_free(dtw)
pa = a.export_plain()
import ctypes
lib = b._lib
parts = []
for r in range(W):
    hr = lib.rb3gpu_shard_handle(sh._s, r)
    n = lib.rb3gpu_get_tot(hr)
    out = np.empty(n, dtype=np.uint8)
    assert lib.rb3gpu_export_plain(hr, out.ctypes.data) == 0
    parts.append(out)
cat = np.concatenate(parts)
print("before the gather: sizes equal %s, identical %s" % (cat.size == pa.size, cat.size == pa.size and bool(np.array_equal(cat, pa))), flush=True)
if cat.size == pa.size and not np.array_equal(cat, pa):
    bad = np.flatnonzero(cat != pa)
    print("  first differences at", bad[:10].tolist(), "of", bad.size, "; bounds", sh.bounds().tolist())
stop = True
if th.is_alive(): th.join()
sh.gather()
pb = b.export_plain()
print("after the gather: identical %s" % bool(pb.size == pa.size and np.array_equal(pb, pa)))
