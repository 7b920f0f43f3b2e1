#!/usr/bin/env python3
"""Synthetic stand-ins for the shapes the mtb star does not cover (SURVEY 8(d) config 5 and VERDICT r2 item 6):

  haplotypes   N haplotypes of one base genome of L bp (0.1 % substitutions each, seed 500 + k), every haplotype cut into
               contigs of random length (mean C bp): long-read assemblies -- several strings of 10^6..10^7 symbols per batch,
               one FASTA file per haplotype (BASELINE configs[4] at 1/30 of a human genome and 4 instead of 579 haplotypes)
  relatives    K close relatives of a short genome (L bp, star phylogeny, 0.1 % substitutions + a few indels), one record each,
               ALL IN ONE FILE: with `-m` chosen for one genome per batch the later rounds see more than 255 matching suffixes
               per interval -- past what the tentative stretches of k_chain track (RB3_TENT_KMAX)

    python tools/gen_family.py haplotypes N L C outdir | relatives K L out.fa
The md5 of the reference's .fmd for the sizes used by the tests is kept in tests/golden/MANIFEST.json ("family"), produced by
tools/make_golden_family.py from oracle/_ref/ropebwt3."""
import os
import sys

import numpy as np

ALPH = np.frombuffer(b"ACGT", dtype=np.uint8)


def _mutate(g, rng, rate):
    g = g.copy()
    n = int(g.size * rate)
    idx = rng.choice(g.size, size=n, replace=False)
    g[idx] = ALPH[(np.searchsorted(ALPH, g[idx]) + rng.integers(1, 4, size=n)) % 4]
    return g


def _fasta(f, name, g):
    s = g.tobytes()
    f.write(b">" + name.encode() + b"\n")
    f.write(b"\n".join(s[i:i + 80] for i in range(0, len(s), 80)) + b"\n")


def haplotypes(N, L, C, out):
    os.makedirs(out, exist_ok=True)
    base = ALPH[np.random.default_rng(7).integers(0, 4, size=L)]
    files = []
    for k in range(N):
        rng = np.random.default_rng(500 + k)
        h = _mutate(base, rng, 0.001)
        cuts = [0]
        while cuts[-1] < L:
            cuts.append(min(L, cuts[-1] + int(rng.integers(C // 4, 2 * C))))
        fn = os.path.join(out, "hap%02d.fa" % k)
        with open(fn, "wb") as f:
            for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
                _fasta(f, "hap%d_ctg%d" % (k, i), h[a:b])
        files.append(fn)
    return files


def relatives(K, L, fn):
    g0 = ALPH[np.random.default_rng(11).integers(0, 4, size=L)]
    with open(fn, "wb") as f:
        for k in range(K):
            rng = np.random.default_rng(900 + k)
            g = _mutate(g0, rng, 0.001)
            p, ln = int(rng.integers(0, L - 600)), int(rng.integers(1, 300))     # one indel each
            g = np.concatenate([g[:p], ALPH[rng.integers(0, 4, size=ln)], g[p:]]) if k % 2 else np.concatenate([g[:p], g[p + ln:]])
            _fasta(f, "rel%d" % k, g)
    return fn


if __name__ == "__main__":
    if sys.argv[1] == "haplotypes":
        print(haplotypes(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]))
    else:
        print(relatives(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]))


# ---- the index of 2^32 symbols and more: N haplotypes of 180 Mbp, contigs of 20-100 Mbp --------------------------
def _mutate_fast(g, rng, rate):
    """substitutions at positions drawn WITH replacement (a permutation of 1.8e8 positions takes longer than everything else)"""
    g = g.copy()
    idx = rng.integers(0, g.size, size=int(g.size * rate))
    g[idx] = ALPH[(np.searchsorted(ALPH, g[idx]) + rng.integers(1, 4, size=idx.size)) % 4]
    return g


def big_base(L, seed=77):
    return ALPH[np.random.default_rng(seed).integers(0, 4, size=L, dtype=np.uint8)]


def big_contigs(base, k, lo=20000000, hi=100000000):
    """haplotype k of the base genome (0.1 % substitutions, seed 7000 + k) as a list of contigs (ASCII arrays) of lo..hi bp"""
    rng = np.random.default_rng(7000 + k)
    h = _mutate_fast(base, rng, 0.001)
    cuts = [0]
    while cuts[-1] < base.size:
        cuts.append(min(base.size, cuts[-1] + int(rng.integers(lo, hi))))
    return [h[a:b] for a, b in zip(cuts[:-1], cuts[1:])]


def big_haplotype_files(N, L, out, lo=20000000, hi=100000000):
    os.makedirs(out, exist_ok=True)
    base = big_base(L)
    files = []
    for k in range(N):
        fn = os.path.join(out, "big%02d.fa" % k)
        if not os.path.exists(fn):
            with open(fn + ".tmp", "wb") as f:
                for i, c in enumerate(big_contigs(base, k, lo, hi)):
                    f.write(b">big%d_ctg%d\n" % (k, i))
                    f.write(c.tobytes() + b"\n")
            os.rename(fn + ".tmp", fn)
        files.append(fn)
    return files
