#!/usr/bin/env python3
"""Differential fuzz of `ropebwt3-amd build` (incl. --gpus N and --gpus N --interval; and, for every third case, `ssa`) against the unmodified reference binary (oracle/_ref/ropebwt3, GPU box):
random FASTA / FASTQ / one-per-line inputs (N, lower case, IUPAC, CRLF, duplicates, homopolymers, short and long
records, several files), random -m / -R / -F / -p / sorter; the .fmd must be byte-identical.
    python tools/fuzz_cli.py [n_cases] [seed0]"""
import hashlib, os, subprocess, sys, gzip
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
amd = os.path.join(ROOT, "ropebwt3_amd", "ropebwt3-amd")
ref = os.path.join(ROOT, "oracle", "_ref", "ropebwt3")
if not os.path.exists(ref):
    sys.exit("no reference binary (oracle/_ref/ropebwt3)")
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
tmp = "/tmp/fuzz_cli"
os.makedirs(tmp, exist_ok=True)
ALPH = np.frombuffer(b"ACGT", dtype=np.uint8)

def seqs(rng):
    kind = rng.integers(0, 5)
    out = []
    if kind == 0:      # genomes
        g = ALPH[rng.integers(0, 4, size=int(rng.integers(500, 40000)))]
        for _ in range(int(rng.integers(1, 8))):
            h = g.copy(); idx = rng.integers(0, len(h), size=max(1, len(h) // 300)); h[idx] = ALPH[rng.integers(0, 4, size=len(idx))]
            out.append(bytes(h))
    elif kind == 1:    # reads with duplicates
        g = ALPH[rng.integers(0, 4, size=3000)]
        L = int(rng.integers(20, 200))
        for _ in range(int(rng.integers(10, 1500))):
            s = int(rng.integers(0, len(g) - L)); out.append(bytes(g[s:s + L]))
        out += out[:int(rng.integers(0, 20))]
    elif kind == 2:    # odd characters
        pool = b"ACGTNacgtnRYKMSWBDHVU-*"
        for _ in range(int(rng.integers(1, 60))):
            out.append(bytes(rng.choice(np.frombuffer(pool, dtype=np.uint8), size=int(rng.integers(1, 300)))))
    elif kind == 3:    # homopolymers and tandem repeats
        for _ in range(int(rng.integers(1, 6))):
            u = bytes(ALPH[rng.integers(0, 4, size=int(rng.integers(1, 12)))])
            out.append(u * int(rng.integers(1, 5000)))
        out.append(b"A" * int(rng.integers(1, 70000)))
    else:              # very short records
        for _ in range(int(rng.integers(1, 400))):
            out.append(bytes(ALPH[rng.integers(0, 4, size=int(rng.integers(1, 6)))]))
    return out

def write(rng, path, recs, fmt):
    eol = b"\r\n" if rng.random() < 0.2 else b"\n"
    if fmt == "line": data = eol.join(recs) + (eol if rng.random() < 0.8 else b"")
    elif fmt == "fa":
        w = int(rng.integers(10, 100)); parts = []
        for i, r in enumerate(recs):
            parts.append(b">s%d c" % i + eol)
            parts += [r[j:j + w] + eol for j in range(0, len(r), w)]
        data = b"".join(parts)
    else: data = b"".join(b"@r%d" % i + eol + r + eol + b"+" + eol + b"I" * len(r) + eol for i, r in enumerate(recs))
    if rng.random() < 0.3:
        path += ".gz"
        with gzip.open(path, "wb") as f: f.write(data)
    else: open(path, "wb").write(data)
    return path

bad = 0
for case in range(ncase):
    rng = np.random.default_rng(seed0 + case)
    fmt = str(rng.choice(["line", "fa", "fq"]))
    files = []
    maxlen = 0
    for fi in range(int(rng.integers(1, 4))):
        recs = seqs(rng)
        maxlen = max([maxlen] + [len(r) for r in recs])
        files.append(write(rng, os.path.join(tmp, "c%d_%d.%s" % (case, fi, "txt" if fmt == "line" else fmt)), recs, fmt))
    flags = (["-L"] if fmt == "line" else []) + ([str(rng.choice(["-R", "-F"]))] if rng.random() < 0.3 else [])
    m = str(rng.choice(["7g", "1", "100", "3k", "50k", "1m"]))
    want = subprocess.run([ref, "build", "-d", "-t4", "-m" + m] + flags + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    host_sort = rng.random() < 0.3
    multi = []
    force_iv = bool(os.environ.get("FUZZ_INTERVAL"))   # (every case that can take it runs with --interval)
    if rng.random() < 0.35 or force_iv:   # several handles (on a one-GPU box all on device 0): slices + tree merge, or ONE index cut into intervals (short records only:
        multi = ["--gpus", str(rng.integers(2, 5))]   # a lock-step round per symbol of the longest record)
        if not host_sort and maxlen < 2000 and (rng.random() < 0.6 or force_iv): multi.append("--interval")
    ours = [amd, "build", "-d", "-m" + m] + flags + (["--host-sort"] if host_sort else []) + (["-p%d" % rng.integers(1, 5)] if rng.random() < 0.5 else []) + \
           (["--rebatch"] if rng.random() < 0.2 and not multi else []) + multi + files
    got = subprocess.run(ours, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    ok = want.returncode == 0 and got.returncode == 0 and want.stdout == got.stdout
    print("case %d: %s %d files m=%s %s -> %s (%d bytes)" % (case, fmt, len(files), m, " ".join(ours[4:-len(files)]), "ok" if ok else "MISMATCH rc=%d/%d" % (want.returncode, got.returncode), len(got.stdout)), flush=True)
    if not ok:
        bad += 1
        print("   ref stderr:", want.stderr.decode()[-300:]); print("   amd stderr:", got.stderr.decode()[-300:])
    if ok and case % 3 == 0:   # and the sampled suffix array of that index, both programs reading the same .fmd
        fmd = os.path.join(tmp, "c%d.fmd" % case)
        open(fmd, "wb").write(got.stdout)
        ss = str(rng.integers(0, 9))
        a = subprocess.run([ref, "ssa", "-t4", "-s" + ss, fmd], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        b = subprocess.run([amd, "ssa", "-s" + ss, fmd], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if not (a.returncode == 0 and b.returncode == 0 and a.stdout == b.stdout):
            bad += 1
            print("   ssa -s%s MISMATCH rc=%d/%d %s" % (ss, a.returncode, b.returncode, b.stderr.decode()[-200:]))
        os.remove(fmd)
    for f in files: os.remove(f)
print("%d cases, %d mismatches" % (ncase, bad))
sys.exit(1 if bad else 0)
