"""rb3gpu_sh_merge at world 1 / 2 / 4 (ranks = threads with a handle each, all on THIS device when the box has one): wall time of the walk per
lock-step round with the rounds as PEER ROUNDS (one kernel per rank and round, states written straight into the owner's receive buffer, the
streams wait for each other's events: rb3gpu_comm_t.stream_barrier) against the rounds driven by the host (read-back + all-gather + all-to-all
per round; tune sh_host_rounds = 1).  Reads of 150 bp, both strands, into an index of 2^26 symbols of a random genome.
    python tools/probe_sh_peer.py [reads ...]            RB3_PROBE_WORLDS=2,4  RB3_PROBE_MODES=0  RB3_PROBE_DEVICES=n
bench.py calls measure() for its aux_interval_peer_rounds object."""
import sys, os, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def measure(sizes, worlds=(1, 2, 4), modes=(0, 1), index_log2=26, reps=3, ndev=1, seed=31):
    """one dict per (reads, world, mode): us of the walk per lock-step round (the slowest rank's), ms per merge, what kind of rounds ran"""
    from ropebwt3_amd import Rb3Gpu, CommGroup, multi
    from tests import util
    rng = np.random.default_rng(seed)
    g = util.random_genome(rng, (1 << index_log2) // 2 - 1)
    t1 = util.make_text([g])
    h0 = Rb3Gpu(verbose=1)
    d1, d1tw = h0.sort_text(t1)
    b1 = h0.dev_download(d1, t1.size)
    h0.dev_free(d1), h0.dev_free(d1tw)
    h0.close()
    res = []
    for n in sizes:
        st = rng.integers(0, len(g) - 150, size=n)
        r = np.stack([g[s:s + 150] for s in st])
        m = rng.random(r.shape) < 0.01
        r[m] = rng.integers(1, 5, size=int(m.sum()), dtype=np.uint8)
        t2 = util.make_text(list(r))
        sent = np.flatnonzero(t2 == 0).astype(np.int64)
        for world in worlds:
            for host_rounds in modes:
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
                    res.append({"reads": int(n), "world": world, "error": repr(errs)[:200]})
                    continue
                rounds = out[0][3]
                kind = "rounds driven by the host" if host_rounds or (world == 1 and sent.size > (1 << 19)) else ("peer rounds" if out[0][4] else "rounds on the device" if world == 1 else "rounds driven by the host (peer rounds NOT taken)")
                res.append({"reads": int(n), "chains": int(sent.size), "symbols": int(t2.size), "world": world, "rounds": int(rounds), "kind": kind,
                            "us_per_round": round(max(o[1] for o in out) / rounds * 1e3, 2), "ms_per_merge": round(max(o[0] for o in out) * 1e3, 3), "rebuild_ms": round(max(o[2] for o in out), 3)})
    return res


if __name__ == "__main__":
    sizes = [int(x) for x in sys.argv[1:]] or [1000, 100000, 1000000]
    worlds = [int(x) for x in os.environ.get("RB3_PROBE_WORLDS", "1,2,4").split(",")]
    modes = [int(x) for x in os.environ.get("RB3_PROBE_MODES", "0,1").split(",")]
    for n in sizes:
        for d in measure([n], worlds, modes, ndev=int(os.environ.get("RB3_PROBE_DEVICES", "1"))):
            if "error" in d:
                print("world %d, %d reads: FAILED %s" % (d["world"], d["reads"], d["error"]), flush=True)
                continue
            print("%8d reads (%9d chains, %10d symbols), world %d, %s: %8.3f ms per merge = %6.3f Gbp/s; walk %7.2f us per round x %d rounds, rebuild %.3f ms" % (
                d["reads"], d["chains"], d["symbols"], d["world"], d["kind"], d["ms_per_merge"], d["symbols"] / d["ms_per_merge"] / 1e6, d["us_per_round"], d["rounds"], d["rebuild_ms"]), flush=True)
