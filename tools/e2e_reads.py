#!/usr/bin/env python3
"""End-to-end `build` in the short-read regime (scaled down): N reads of 150 bp sampled
from a random genome at ~30x with 1 % substitution errors, one FASTA, batches cut by -m as the reference does.
Runs ropebwt3-amd and (unless --no-ref) the reference, compares the .fmd byte for byte.
    python tools/e2e_reads.py N_READS [BATCH e.g. 70m] [outdir] [--no-ref]"""
import hashlib, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NOREF = "--no-ref" in sys.argv
if NOREF: sys.argv.remove("--no-ref")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
M = sys.argv[2] if len(sys.argv) > 2 else "70m"
out = sys.argv[3] if len(sys.argv) > 3 else "/tmp/e2e_reads"
os.makedirs(out, exist_ok=True)
rng = np.random.default_rng(4)
ALPH = np.frombuffer(b"ACGT", dtype=np.uint8)
G = ALPH[rng.integers(0, 4, size=max(1000, N * 150 // 30))]
fn = os.path.join(out, "reads.fa")
t = time.time()
with open(fn, "wb") as f:
    for b0 in range(0, N, 100000):
        n = min(100000, N - b0)
        st = rng.integers(0, len(G) - 150, size=n)
        r = G[st[:, None] + np.arange(150)[None, :]]
        m = rng.random(r.shape) < 0.01
        r[m] = ALPH[rng.integers(0, 4, size=int(m.sum()))]
        lines = np.empty((n, 151), dtype=np.uint8); lines[:, :150] = r; lines[:, 150] = 10
        hdr = [b">r%d\n" % (b0 + i) for i in range(n)]
        body = lines.tobytes()
        f.write(b"".join(h + body[i * 151:(i + 1) * 151] for i, h in enumerate(hdr)))
print("generated %d reads (%d symbols both strands) in %.1f s" % (N, N * 302, time.time() - t), flush=True)
def run(name, cmd):
    t = time.time()
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    dt = time.time() - t
    print("%-30s %7.1f s  rc=%d  md5=%s  bytes=%d" % (name, dt, r.returncode, hashlib.md5(r.stdout).hexdigest(), len(r.stdout)), flush=True)
    err = r.stderr.decode()
    for l in err.splitlines():
        if "GPU merge path" in l or "Real time" in l or "ERROR" in l or "E::" in l: print("    " + l)
    return r.stdout, err
amd = os.path.join(ROOT, "ropebwt3_amd", "ropebwt3-amd")
ref = os.path.join(ROOT, "oracle", "_ref", "ropebwt3")
a, ea = run("amd build -m%s" % M, [amd, "build", "-d", "-m" + M, fn])
for l in ea.splitlines():
    if "GPU suffix sorting" in l: print("    " + l)
print("    merge rounds:", ea.count("merged the partial BWT"))
a2, _ = run("amd build --host-sort -m%s -p8" % M, [amd, "build", "-d", "--host-sort", "-m" + M, "-p8", fn])
print("host-sort identical:", a == a2)
if os.path.exists(ref) and not NOREF:
    import re
    b, eb = run("reference build -m%s -t64" % M, [ref, "build", "-d", "-m" + M, "-t%d" % min(64, os.cpu_count() or 8), fn])
    print("IDENTICAL to reference:", a == b)
    tot, last = 0.0, None
    for l in eb.splitlines():
        m = re.match(r"\[M::\w+::([0-9.]+)\*", l)
        if not m: continue
        if "constructed partial BWT" in l: last = float(m.group(1))
        elif "inserted" in l and last is not None: tot += float(m.group(1)) - last; last = None
    print("reference merge-only seconds (sum over rounds): %.2f" % tot)
