#!/usr/bin/env python3
"""Synthetic stand-in for BASELINE configs[3] (30x human 150 bp reads, -m7g), scaled: N reads of 150 bp sampled uniformly
from a random genome at ~30x with 1 % substitution errors (seed 4), one read per line (`build -L`).  N = 10,000,000 gives
3.02 G symbols with both strands: more than 2^31, i.e. a single `-m7g` batch that the GPU suffix sorter cannot take whole.
The md5 of the reference's .fmd for it is kept in tests/golden/MANIFEST.json ("reads_m7g").
    python tools/gen_reads.py N out.txt"""
import sys
import numpy as np

ALPH = np.frombuffer(b"ACGT", dtype=np.uint8)


def generate(N, fn, seed=4):
    rng = np.random.default_rng(seed)
    G = ALPH[rng.integers(0, 4, size=max(1000, N * 150 // 30))]
    with open(fn, "wb") as f:
        for b0 in range(0, N, 200000):
            n = min(200000, N - b0)
            st = rng.integers(0, len(G) - 150, size=n)
            r = G[st[:, None] + np.arange(150)[None, :]]
            m = rng.random(r.shape) < 0.01
            r[m] = ALPH[rng.integers(0, 4, size=int(m.sum()))]
            lines = np.empty((n, 151), dtype=np.uint8)
            lines[:, :150] = r
            lines[:, 150] = 10
            f.write(lines.tobytes())
    return fn


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    print(generate(N, sys.argv[2] if len(sys.argv) > 2 else "/tmp/reads.txt"))
