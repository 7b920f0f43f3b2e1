#!/usr/bin/env python3
"""Generate tests/golden/: inputs + outputs of the UNMODIFIED reference (oracle/_ref/ropebwt3,
built by oracle/Makefile from /root/reference).  Run in the build container only; the fixtures
(data, not code) are committed so that the GPU box and CI never need the reference.

    python tools/make_golden.py

Every fixture is a pair (input file, expected output); MANIFEST.json lists the command line the
reference was run with, the md5 of its .fmd and, for small cases, the plain BWT.
"""
import gzip
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "ropebwt3")
OUT = os.path.join(ROOT, "tests", "golden")
ALPH = np.frombuffer(b"ACGT", dtype=np.uint8)


def ref(args, stdin=None):
    r = subprocess.run([REF] + args, input=stdin, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, (args, r.stderr.decode()[-500:])
    return r.stdout


def dna(rng, n):
    return ALPH[rng.integers(0, 4, size=n)]


def mutate(rng, g, rate):
    g = g.copy()
    k = int(len(g) * rate)
    idx = rng.choice(len(g), size=k, replace=False)
    g[idx] = ALPH[(np.searchsorted(ALPH, g[idx]) + rng.integers(1, 4, size=k)) % 4]
    return g


def fasta(seqs, width=70):
    out = []
    for i, s in enumerate(seqs):
        out.append(b">seq%d some comment\n" % i)
        s = bytes(s)
        for j in range(0, len(s), width):
            out.append(s[j:j + width] + b"\n")
    return b"".join(out)


def main():
    os.makedirs(OUT, exist_ok=True)
    man = {}
    rng = np.random.default_rng(20260927)

    def add(name, data, line_mode, extra_flags=(), m_variants=("7g",), keep_plain=False, gz=True, files=None):
        """write the input, run the reference with several -m, make sure all agree, record"""
        if files is None:
            fn = name + (".txt" if line_mode else ".fa")
            files = [(fn, data)]
        paths = []
        for fn, d in files:
            if gz:
                fn += ".gz"
                with gzip.GzipFile(os.path.join(OUT, fn), "wb", mtime=0) as f:
                    f.write(d)
            else:
                open(os.path.join(OUT, fn), "wb").write(d)
            paths.append(fn)
        flags = (["-L"] if line_mode else []) + list(extra_flags)
        fmds = set()
        for m in m_variants:
            fmd = ref(["build"] + flags + ["-m" + m, "-t4", "-d"] + [os.path.join(OUT, p) for p in paths])
            fmds.add(fmd)
        assert len(fmds) == 1, "%s: the reference's .fmd depends on -m?!" % name
        fmd = fmds.pop()
        with open(os.path.join(OUT, name + ".fmd"), "wb") as f:
            f.write(fmd)
        ent = {"inputs": paths, "flags": flags, "m_variants": list(m_variants), "fmd": name + ".fmd",
               "fmd_md5": hashlib.md5(fmd).hexdigest(), "fmd_bytes": len(fmd)}
        plain = ref(["build"] + flags + ["-m" + m_variants[-1], "-t4"] + [os.path.join(OUT, p) for p in paths])
        ent["plain_md5"] = hashlib.md5(plain).hexdigest()
        ent["n_symbols"] = len(plain) - 1
        if keep_plain:
            with gzip.GzipFile(os.path.join(OUT, name + ".bwt.gz"), "wb", mtime=0) as f:
                f.write(plain)
            ent["plain"] = name + ".bwt.gz"
        if len(plain) < 200:
            ent["plain_text"] = plain.decode().strip()
        man[name] = ent
        print("%-12s %8d symbols  fmd %7d B  md5 %s" % (name, ent["n_symbols"], len(fmd), ent["fmd_md5"]))

    # K1-K3: two toy strings
    add("k2_fwd", b"AGG\nAGC\n", True, ["-R"], ("7g", "1"), keep_plain=True, gz=False)
    add("k3_both", b"AGG\nAGC\n", True, [], ("7g", "1"), keep_plain=True, gz=False)
    # K4: the README 31-mer
    add("k4_readme", b"TGAACTCTACACAACATATTTTGTCACCAAG\n", True, [], ("7g",), keep_plain=True, gz=False)
    # edge: duplicates, N, lower case
    add("edge_dups", b"ACG\nACG\nTTT\nACG\nNNA\nacgtn\n", True, [], ("7g", "9", "1"), keep_plain=True, gz=False)
    # IUPAC / odd characters, CRLF line ends, no trailing newline
    add("edge_chars", b"ACGTRYKM\r\nNNNN\r\nacgtu-*\r\nA\r\nTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTT", True, [], ("7g", "20", "1"), keep_plain=True, gz=False)
    # similar genomes, multi-line FASTA with comments
    g = dna(rng, 20000)
    gen = [mutate(rng, g, 0.005) for _ in range(12)]
    add("genomes12", fasta(gen), False, [], ("7g", "100k", "45k"), keep_plain=True)
    # same genomes as 3 separate files: batches never span files in the reference
    add("genomes12_files", None, False, [], ("7g", "45k"), keep_plain=False,
        files=[("genomes12_part%d.fa" % i, fasta(gen[4 * i:4 * i + 4])) for i in range(3)])
    assert man["genomes12_files"]["fmd_md5"] == man["genomes12"]["fmd_md5"]
    # short reads with exact duplicates and reverse-complement palindromes, FASTQ
    g2 = dna(rng, 30000)
    reads = []
    for _ in range(3000):
        s = int(rng.integers(0, len(g2) - 100))
        r = g2[s:s + 100].copy()
        m = rng.random(100) < 0.01
        r[m] = ALPH[rng.integers(0, 4, size=int(m.sum()))]
        reads.append(r)
    reads += reads[:50]
    comp = {65: 84, 67: 71, 71: 67, 84: 65}
    half = dna(rng, 20)
    pal = np.concatenate([half, np.array([comp[x] for x in half[::-1]], dtype=np.uint8)])
    reads += [pal, pal]
    fq = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, bytes(r), b"I" * len(r)) for i, r in enumerate(reads))
    add("reads_fq", fq, False, [], ("7g", "200k", "50k"), keep_plain=True)
    # forward strand only, line input
    add("reads_fwd", b"".join(bytes(r) + b"\n" for r in reads[:1000]), True, ["-R"], ("7g", "30k"), keep_plain=True)
    # reverse strand only
    add("reads_rev", b"".join(bytes(r) + b"\n" for r in reads[:500]), True, ["-F"], ("7g", "30k"), keep_plain=False)
    # many copies of one 500-mer: FMD blocks with >= 0x4000 symbols -> 32-bit block headers
    mer = bytes(dna(rng, 500))
    add("copies3000", (mer + b"\n") * 3000, True, [], ("7g", "400k"), keep_plain=False)
    # long homopolymers (runs > 2^19 -> 8-byte FMR codes, long delta codes) and an all-N sequence
    add("longruns", b"A" * 700000 + b"\n" + b"ACGT" * 1000 + b"\n" + b"N" * 3000 + b"\n" + b"T" * 300000 + b"C" * 5 + b"\n", True, [], ("7g", "600k"), keep_plain=False)

    # -i resume: FMR and FMD checkpoints of the first 6 genomes, then the rest
    first = os.path.join(OUT, "genomes12_first6.fa")
    rest = os.path.join(OUT, "genomes12_rest6.fa")
    open(first, "wb").write(fasta(gen[:6]))
    open(rest, "wb").write(fasta(gen[6:]))
    fmr = ref(["build", "-t4", "-b", first])
    open(os.path.join(OUT, "genomes12_first6.fmr"), "wb").write(fmr)
    fmd6 = ref(["build", "-t4", "-d", first])
    open(os.path.join(OUT, "genomes12_first6.fmd"), "wb").write(fmd6)
    for src in ("genomes12_first6.fmr", "genomes12_first6.fmd"):
        out = ref(["build", "-t4", "-d", "-i", os.path.join(OUT, src), rest])
        assert hashlib.md5(out).hexdigest() == man["genomes12"]["fmd_md5"]
    man["resume"] = {"first": "genomes12_first6.fa", "rest": "genomes12_rest6.fa", "fmr": "genomes12_first6.fmr",
                     "fmd": "genomes12_first6.fmd", "expect": "genomes12"}
    os.remove(first)
    with gzip.GzipFile(first + ".gz", "wb", mtime=0) as f:
        f.write(fasta(gen[:6]))
    os.remove(rest)
    with gzip.GzipFile(rest + ".gz", "wb", mtime=0) as f:
        f.write(fasta(gen[6:]))
    man["resume"]["first"] += ".gz"
    man["resume"]["rest"] += ".gz"

    # sampled suffix arrays of the fixtures' indexes (`ropebwt3 ssa`, ssa.c): md5 of the .ssa file for
    # several sample rates; the file itself for two of them
    for name, ent in sorted(man.items()):
        if "fmd" not in ent or name == "resume":
            continue
        ent["ssa_md5"] = {}
        for ss in (0, 3, 8):
            out = ref(["ssa", "-t4", "-s%d" % ss, os.path.join(OUT, ent["fmd"])])
            ent["ssa_md5"][str(ss)] = hashlib.md5(out).hexdigest()
            if (name, ss) in (("k3_both", 0), ("genomes12", 8), ("reads_fwd", 3)):
                fn = "%s.s%d.ssa" % (name, ss)
                open(os.path.join(OUT, fn), "wb").write(out)
                ent["ssa_file"] = {"shift": ss, "file": fn}

    json.dump(man, open(os.path.join(OUT, "MANIFEST.json"), "w"), indent=1, sort_keys=True)
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("total fixture bytes:", tot)


if __name__ == "__main__":
    sys.exit(main())
