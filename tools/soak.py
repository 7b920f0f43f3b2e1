"""Randomised soak of the merge path against the oracle (GPU box):  python tools/soak.py [n_cases] [seed0]
Families of relatives with substitutions, indels, exact duplicates, homopolymer stretches and tandem repeats,
merged one by one through the single-sync walker path (and every few cases through the automatic split);
the merged BWT, the run export and the sampled suffix array are compared with the oracle's."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import util
from ropebwt3_amd import Rb3Gpu, host

ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
orc = util.Oracle()
fb = 0
t0 = time.time()
for case in range(ncase):
    rng = np.random.default_rng(seed0 + case)
    L = int(rng.integers(3000, 60000))
    base = util.random_genome(rng, L)
    if rng.random() < 0.5:    # a tandem repeat and a homopolymer inside
        unit = util.random_genome(rng, int(rng.integers(2, 40)))
        base = np.concatenate([base[:L // 3], np.tile(unit, int(rng.integers(5, 200))), base[L // 3:2 * L // 3],
                               np.full(int(rng.integers(10, 3000)), int(rng.integers(1, 5)), dtype=np.uint8), base[2 * L // 3:]])
    nrel = int(rng.integers(2, 70))
    rel = [base]
    for i in range(nrel - 1):
        src = rel[int(rng.integers(0, len(rel)))]
        g = src.copy() if rng.random() < 0.15 else util.mutate(rng, src, float(rng.choice([0.0002, 0.001, 0.004, 0.02])))
        if rng.random() < 0.3:   # an indel
            p = int(rng.integers(0, len(g) - 10)); ln = int(rng.integers(1, 300))
            g = np.concatenate([g[:p], util.random_genome(rng, ln), g[p:]]) if rng.random() < 0.5 else np.concatenate([g[:p], g[min(len(g), p + ln):]])
        rel.append(g)
    both = bool(rng.random() < 0.7)
    step = int(rng.choice([128, 200, 384, 1000]))
    per = int(rng.choice([1, 1, 2, 5]))          # genomes per batch
    h = Rb3Gpu(verbose=1, hooks=bool(os.environ.get("RB3GPU_TEXT_MODE")))   # (test hooks need the test build of the library)
    cur = None
    for i in range(0, nrel, per):
        t = util.make_text(rel[i:i + per], True, both)
        if case % 4 == 3 and not os.environ.get("SOAK_TEXT"):
            b = host.build_bwt(t)
            if cur is None: h.from_plain(b)
            else: h.merge_plain(b)
        elif (case % 4 == 2 and case % 8 != 6) or os.environ.get("SOAK_TEXT"):   # suffix-sorted on the GPU, merged through its text-order words (the CLI's default path)
            d, dtw, dsa = h.sort_text_sa(t)       # (with the suffix array: RB3GPU_TREC=1 then leaves the records in text order)
            b = h.dev_download(d, t.size)
            if not np.array_equal(b, host.build_bwt(t.copy())):
                print("case %d: GPU suffix sorter MISMATCH" % case); sys.exit(1)
            if cur is None: h._chk(h._lib.rb3gpu_from_plain_dev(h._h, t.size, d), "from_plain_dev")
            else: h.merge_text_dev(d, dtw, t.size, host.walkers_text(t, step), commit=True, d_sa=dsa)
            h.dev_free(d); h.dev_free(dtw); h.dev_free(dsa)
        elif case % 4 == 2:   # the batch is suffix-sorted on the GPU as well; the BWT never leaves the device
            d, ck = h.bwt_from_text(t, step)
            b = h.dev_download(d, t.size)
            if not np.array_equal(b, host.build_bwt(t.copy())):
                print("case %d: GPU suffix sorter MISMATCH" % case); sys.exit(1)
            if cur is None: h._chk(h._lib.rb3gpu_from_plain_dev(h._h, t.size, d), "from_plain_dev")
            else:
                w = host.walkers_from_ckrow(t, step, ck)
                h.merge_plain_dev_walkers(d, t.size, w, commit=True)
            h.dev_free(d)
        else:
            b, w = host.build_bwt_walkers(t, step)
            if cur is None: h.from_plain(b)
            else: h.merge_plain_walkers(b, w)
        cur = b if cur is None else orc.merge(cur, b)
    got = h.export_plain()
    ok = np.array_equal(got, cur)
    runs = h.export_runs()
    ok_runs = runs == orc.runs(cur)
    ss = int(rng.choice([0, 3, 6]))
    ms, r2i, ssa = h.ssa_gen(ss)
    want = orc.ssa_gen(cur, ss)
    ok_ssa = ms == want[0] and np.array_equal(r2i, want[1]) and np.array_equal(ssa, want[2])
    st = h.stats()
    fb += st["n_fallbacks"]
    h.close()
    print("case %d: L=%d nrel=%d both=%d step=%d per=%d n=%d  merge %s runs %s ssa %s  fallbacks %d" % (case, len(base), nrel, both, step, per, cur.size,
          "ok" if ok else "MISMATCH", "ok" if ok_runs else "MISMATCH", "ok" if ok_ssa else "MISMATCH", st["n_fallbacks"]), flush=True)
    if not (ok and ok_runs and ok_ssa):
        sys.exit(1)
print("all %d cases ok, %d fallbacks, %.1f s" % (ncase, fb, time.time() - t0))
