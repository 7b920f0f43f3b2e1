import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import util
from ropebwt3_amd import Rb3Gpu, host
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
rate = float(sys.argv[2]) if len(sys.argv) > 2 else 0.002
sl = int(sys.argv[3]) if len(sys.argv) > 3 else 8
g0 = util.random_genome(np.random.default_rng(1), n); g1 = util.mutate(np.random.default_rng(2), g0, rate)
b1 = host.build_bwt(util.make_text([g0])); b2 = host.build_bwt(util.make_text([g1]))
orc = util.Oracle(); rb, _ = orc.mg_rank(b1, b2, 8)
h = Rb3Gpu(split_log2=sl, verbose=1); h.from_plain(b1); d = h.dev_upload(b2)
try:
    h.merge_plain_dev(d, b2.size, commit=False)
    print("merge ok")
except Exception as e:
    print("ERR", e)
# raw pos
h.mg_begin(d, b2.size)
try:
    h._chk(h._lib.rb3gpu_mg_walk(h._h, 0, None, -1, None), "walk")  # public walk = non-tentative
except Exception as e: print("walk err", e)
p, ln = h.mg_pos_ptr(); pos = np.empty(ln, dtype=np.int64); h._chk(h._lib.rb3gpu_dev_download(h._h, pos.ctypes.data, p, ln * 8), "dl")
print("non-tent equal:", np.array_equal(pos, rb >> 6))
