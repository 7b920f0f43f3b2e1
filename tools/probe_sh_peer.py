"""rb3gpu_sh_merge at world 1 / 2 / 4 (ranks = threads with a handle each, all on THIS device when the box has one): wall time of the walk per
lock-step round with the rounds as PEER ROUNDS (one kernel per rank and round, states written straight into the owner's receive buffer, the
streams wait for each other's events: rb3gpu_comm_t.stream_barrier) against the rounds driven by the host (read-back + all-gather + all-to-all
per round; tune sh_host_rounds = 1).  Reads of 150 bp, both strands, into an index of 2^26 symbols of a random genome.
    python tools/probe_sh_peer.py [reads ...]"""
import sys, os, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ropebwt3_amd import Rb3Gpu, CommGroup, host, multi
from tests import util

sizes = [int(x) for x in sys.argv[1:]] or [1000, 100000, 1000000]
rng = np.random.default_rng(31)
g = util.random_genome(rng, (1 << 26) // 2 - 1)
t1 = util.make_text([g])
h0 = Rb3Gpu(verbose=1)
d1, d1tw = h0.sort_text(t1)
b1 = h0.dev_download(d1, t1.size)
h0.dev_free(d1), h0.dev_free(d1tw)
h0.close()
ndev = int(os.environ.get("RB3_PROBE_DEVICES", "1"))

for n in sizes:
    st = rng.integers(0, len(g) - 150, size=n)
    r = np.stack([g[s:s + 150] for s in st])
    m = rng.random(r.shape) < 0.01
    r[m] = rng.integers(1, 5, size=int(m.sum()), dtype=np.uint8)
    t2 = util.make_text(list(r))
    sent = np.flatnonzero(t2 == 0).astype(np.int64)
    for world in [int(x) for x in os.environ.get("RB3_PROBE_WORLDS", "1,2,4").split(",")]:
        for host_rounds in [int(x) for x in os.environ.get("RB3_PROBE_MODES", "0,1").split(",")]:
            grp = CommGroup(world)
            bounds0 = multi.interval_bounds(b1.size, world)
            out, errs = [None] * world, []
            bar = threading.Barrier(world)

            def run(rank):
                try:
                    h = Rb3Gpu(device=rank % ndev, verbose=1)
                    if host_rounds:
                        h.tune("sh_host_rounds", 1)
                    comm = grp.comm(rank, h)
                    h.from_plain(b1[bounds0[rank]:bounds0[rank + 1]])
                    d2, d2tw = h.sort_text(t2)
                    h.sh_merge(comm, bounds0, d2, d2tw, t2.size, sent, commit=False)
                    h.stats_reset()
                    bar.wait()
                    t = time.perf_counter()
                    reps = 3
                    for _ in range(reps):
                        _, rounds = h.sh_merge(comm, bounds0, d2, d2tw, t2.size, sent, commit=False)
                    dt = (time.perf_counter() - t) / reps
                    s = h.stats()
                    out[rank] = (dt, s["ms_rank"] / reps, s["ms_build"] / reps, rounds, s["n_peer_rounds"] // reps)
                    h.dev_free(d2), h.dev_free(d2tw)
                    h.close()
                except BaseException as e:
                    errs.append((rank, repr(e)))
                    grp.abort()
                    try:
                        bar.abort()
                    except Exception:
                        pass

            th = [threading.Thread(target=run, args=(q,)) for q in range(world)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            grp.close()
            if errs:
                print("world %d, %d reads: FAILED %r" % (world, n, errs), flush=True)
                continue
            dt = max(o[0] for o in out)
            walk = max(o[1] for o in out)
            rounds = out[0][3]
            print("%8d reads (%9d chains, %10d symbols), world %d, %s: %8.3f ms per merge = %6.3f Gbp/s; walk %7.2f us per round x %d rounds, rebuild %.3f ms%s" % (
                n, sent.size, t2.size, world, "rounds driven by the host" if host_rounds else ("peer rounds" if out[0][4] else "rounds on the device"),
                dt * 1e3, t2.size / dt / 1e9, walk / rounds * 1e3, rounds, max(o[2] for o in out), "" if host_rounds or world == 1 or out[0][4] == rounds else "  [peer rounds NOT taken]"), flush=True)
