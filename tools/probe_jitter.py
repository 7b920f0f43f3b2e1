import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from tests import util
from ropebwt3_amd import Rb3Gpu, host
g0 = util.random_genome(np.random.default_rng(1), 4400000); g1 = util.mutate(np.random.default_rng(2), g0, 0.001)
b1 = host.build_bwt(util.make_text([g0])); t2 = util.make_text([g1])
h = Rb3Gpu(verbose=1); h.from_plain(b1)
d, ck = h.bwt_from_text(t2, 16)
b2 = h.dev_download(d, t2.size)
ends = np.flatnonzero(t2 == 0)
rng = np.random.default_rng(7)
def mk(jit, W=384):
    w = []; b = 0
    for j, e in enumerate(ends):
        ps = np.arange(b // W + 1, e // W + 1) * W
        ps = ps + (rng.exponential(jit, size=ps.size) if jit else 0)
        ps = (ps.astype(np.int64) // 16) * 16
        ps = ps[(ps > b) & (e - ps >= 128)]
        ps = np.unique(ps)
        prev = -1
        for p in ps:
            w.append((ck[p // 16], -1, (1 << 62) if prev < 0 else p - prev, 0)); prev = p
        w.append((j, -2, (1 << 62) if prev < 0 else e - prev, 0)); b = e + 1
    return np.array(w, dtype=np.int64)
for jit, W in ((0, 384), (16, 384), (32, 384), (64, 384), (0, 320), (16, 320), (0, 256), (16, 256), (32, 256), (0, 192), (16, 192)):
    w = mk(jit, W)
    for i in range(3): h.merge_plain_dev_walkers(d, t2.size, w, commit=False)
    h.stats_reset()
    for i in range(10): h.merge_plain_dev_walkers(d, t2.size, w, commit=False)
    st = h.stats()
    ns = w[:, 2]; ns = ns[ns < (1 << 60)]
    print("W %d jitter %2d: %d walkers, gaps min %d mean %.0f max %d; k_chain %.3f ms, steps %.2fM fb %d" % (W, jit, len(w), ns.min(), ns.mean(), ns.max(), st["ms_chain"] / 10, st["n_lf_steps"] / 10 / 1e6, st["n_fallbacks"]))
for i in range(3): h.merge_plain_dev(d, t2.size, commit=False)
h.stats_reset()
for i in range(10): h.merge_plain_dev(d, t2.size, commit=False)
st = h.stats()
print("BWT-only entry point (device-made list): k_chain %.3f ms, rank phase %.3f ms, steps %.2fM fb %d" % (st["ms_chain"] / 10, st["ms_rank"] / 10, st["n_lf_steps"] / 10 / 1e6, st["n_fallbacks"]))
