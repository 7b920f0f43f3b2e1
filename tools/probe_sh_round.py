"""rb3gpu_sh_merge at world 1: wall time of the walk per lock-step round against the number of chains (reads of 150 bp, both strands,
into an index of 2^26 symbols of a random genome).   python tools/probe_sh_round.py [reads ...]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ropebwt3_amd import Rb3Gpu, CommGroup, multi
from tests import util
sizes = [int(x) for x in sys.argv[1:]] or [1000, 10000, 100000, 500000]
h = Rb3Gpu(verbose=1)
rng = np.random.default_rng(31)
g = util.random_genome(rng, (1 << 26) // 2 - 1)
t1 = util.make_text([g])
d1, d1tw = h.sort_text(t1)
h.from_plain_dev(d1, t1.size)
h.dev_free(d1), h.dev_free(d1tw)
comm = CommGroup(1).comm(0, h)
for n in sizes:
    st = rng.integers(0, len(g) - 150, size=n)
    r = np.stack([g[s:s + 150] for s in st])
    m = rng.random(r.shape) < 0.01
    r[m] = rng.integers(1, 5, size=int(m.sum()), dtype=np.uint8)
    t2 = util.make_text(list(r))
    d2, d2tw = h.sort_text(t2)
    sent = np.flatnonzero(t2 == 0).astype(np.int64)
    bounds = multi.interval_bounds(h.get_tot(), 1)
    h.sh_merge(comm, bounds, d2, d2tw, t2.size, sent, commit=False)
    h.stats_reset()
    t = time.perf_counter()
    reps = 3
    for _ in range(reps):
        _, rounds = h.sh_merge(comm, bounds, d2, d2tw, t2.size, sent, commit=False)
    dt = (time.perf_counter() - t) / reps
    s = h.stats()
    print("%8d reads (%9d chains, %10d symbols): %8.3f ms per merge = %6.3f Gbp/s; walk %7.2f us per round x %d rounds (%.2f G steps/s), rebuild %.3f ms" % (
        n, sent.size, t2.size, dt * 1e3, t2.size / dt / 1e9, s["ms_rank"] / reps / rounds * 1e3, rounds, t2.size * reps / s["ms_rank"] / 1e6, s["ms_build"] / reps), flush=True)
    h.dev_free(d2), h.dev_free(d2tw)
h.close()
