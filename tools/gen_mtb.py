#!/usr/bin/env python3
"""Synthetic stand-in for BASELINE configs[2] (mtb152): K genomes of L bp in a star
phylogeny -- each is the ancestor G0 (uniform random ACGT, seed 1) with 0.1 % substitutions and 10 indels of
up to 1 kb (seed 100 + k) --, one FASTA file per genome, as the reference is run on it (one file per batch).
The md5 of the reference's .fmd for (K, L) = (152, 4400000) and smaller prefixes is kept in
tests/golden/MANIFEST.json ("mtb_star"), produced by tools/make_golden_mtb.py from oracle/_ref/ropebwt3.

    python tools/gen_mtb.py K L outdir        # writes outdir/g000.fa ...
"""
import os
import sys

import numpy as np

ALPH = np.frombuffer(b"ACGT", dtype=np.uint8)


def ancestor(L):
    return ALPH[np.random.default_rng(1).integers(0, 4, size=L)]


def genome(k, g0):
    """genome k of the star: ASCII bytes"""
    L = g0.size
    rng = np.random.default_rng(100 + k)
    g = g0.copy()
    idx = rng.choice(L, size=L // 1000, replace=False)
    g[idx] = ALPH[(np.searchsorted(ALPH, g[idx]) + rng.integers(1, 4, size=idx.size)) % 4]
    parts, last = [], 0
    for p in sorted(rng.integers(0, L - 2000, size=10)):      # 10 indels <= 1 kb
        if p < last:
            continue
        parts.append(g[last:p])
        ln = int(rng.integers(1, 1000))
        if rng.random() < 0.5:
            parts.append(ALPH[rng.integers(0, 4, size=ln)])
            last = p
        else:
            last = p + ln
    parts.append(g[last:])
    return np.concatenate(parts)


def write_fasta(fn, name, g):
    with open(fn, "wb") as f:
        f.write(b">" + name.encode() + b"\n")
        s = g.tobytes()
        f.write(b"\n".join(s[i:i + 80] for i in range(0, len(s), 80)) + b"\n")


def generate(K, L, out, first=0):
    """write genomes first..K-1 (files that exist with the right name are regenerated: cheap and deterministic)"""
    os.makedirs(out, exist_ok=True)
    g0 = ancestor(L)
    files = []
    for k in range(K):
        fn = os.path.join(out, "g%03d.fa" % k)
        if k >= first:
            write_fasta(fn, "g%d" % k, genome(k, g0))
        files.append(fn)
    return files


if __name__ == "__main__":
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 4400000
    out = sys.argv[3] if len(sys.argv) > 3 else "/tmp/mtb_star"
    t = generate(K, L, out)
    print("generated %d genomes of ~%d bp in %s" % (len(t), L, out))
