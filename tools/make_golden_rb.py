#!/usr/bin/env python3
"""Intermediate vectors of the reference: rb[] after rb3_mg_rank_plain (fm-index.c:202-225) for seeded inputs,
computed by the UNMODIFIED reference's own function in oracle/_ref/librb3ref.so (it exports rb3_mg_rank_plain: no instrumented
build is needed) and stored as numbers in tests/golden/rb_vectors.npz.  The inputs are regenerated from the seeds by the tests.
    python tools/make_golden_rb.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import util


def cases():
    """name -> (lines of the indexed text, lines of the batch, both strands?)"""
    out = {"k2": (["AGG"], ["AGC"], False), "k3": (["AGG"], ["AGC"], True)}
    rng = np.random.default_rng(4242)
    g0 = util.random_genome(rng, 3000)
    fam = [g0] + [util.mutate(rng, g0, 0.004) for _ in range(5)]
    S = lambda g: "".join("$ACGTN"[x] for x in g)
    out["family"] = ([S(g) for g in fam[:4]], [S(g) for g in fam[4:]] + [S(fam[1])], True)           # incl. an exact duplicate of an indexed genome
    reads = [S(r) for r in util.reads_from(rng, g0, 60, 70, err=0.02)]
    out["reads"] = ([S(g0)] + reads[:20], reads[20:] + [reads[3], "N" * 30, "ACGT" * 10], True)      # duplicates, all-N, a tandem repeat
    return out


if __name__ == "__main__":
    ref, orc = util.Reference(), util.Oracle()
    vec = {}
    for name, (l1, l2, both) in cases().items():
        b1 = ref.bwt(orc.text(l1, True, both))
        b2 = ref.bwt(orc.text(l2, True, both))
        rb, acc2 = ref.mg_rank(b1, b2)
        vec[name + "_rb"], vec[name + "_acc2"] = rb, acc2
        print(name, b1.size, b2.size, rb[:6] >> 6)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "rb_vectors.npz"), **vec)
