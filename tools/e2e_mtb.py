#!/usr/bin/env python3
"""End-to-end `build` on a synthetic mtb-like collection (scaled): K genomes of
L bp in a star phylogeny (0.1 % substitutions + a few indels each), one FASTA per genome.
Runs the reference binary (oracle/_ref/ropebwt3, if present) and ropebwt3-amd, compares the .fmd
byte for byte and prints the timing lines.   python tools/e2e_mtb.py K L [outdir]"""
import hashlib, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
L = int(sys.argv[2]) if len(sys.argv) > 2 else 4400000
NOREF = "--no-ref" in sys.argv
if NOREF: sys.argv.remove("--no-ref")
SSA = "--ssa" in sys.argv
if SSA: sys.argv.remove("--ssa")
out = sys.argv[3] if len(sys.argv) > 3 else "/tmp/e2e_mtb"
os.makedirs(out, exist_ok=True)
ALPH = np.frombuffer(b"ACGT", dtype=np.uint8)
g0 = ALPH[np.random.default_rng(1).integers(0, 4, size=L)]
files = []
for k in range(K):
    rng = np.random.default_rng(100 + k)
    g = g0.copy()
    idx = rng.choice(L, size=L // 1000, replace=False)
    g[idx] = ALPH[(np.searchsorted(ALPH, g[idx]) + rng.integers(1, 4, size=idx.size)) % 4]
    parts, last = [], 0
    for p in sorted(rng.integers(0, L - 2000, size=10)):      # 10 indels <= 1 kb
        if p < last: continue
        parts.append(g[last:p]); ln = int(rng.integers(1, 1000))
        if rng.random() < 0.5: parts.append(ALPH[rng.integers(0, 4, size=ln)]); last = p
        else: last = p + ln
    parts.append(g[last:])
    g = np.concatenate(parts)
    fn = os.path.join(out, "g%03d.fa" % k)
    with open(fn, "wb") as f:
        f.write(b">g%d\n" % k)
        s = g.tobytes()
        f.write(b"\n".join(s[i:i + 80] for i in range(0, len(s), 80)) + b"\n")
    files.append(fn)
print("generated %d genomes of ~%d bp" % (K, L), flush=True)
def run(name, cmd):
    t = time.time()
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    dt = time.time() - t
    err = r.stderr.decode()
    print("%-28s %7.1f s  rc=%d  md5=%s  bytes=%d" % (name, dt, r.returncode, hashlib.md5(r.stdout).hexdigest(), len(r.stdout)), flush=True)
    for l in err.splitlines():
        if "GPU merge path" in l or "Real time" in l: print("    " + l)
    return r.stdout, err
amd = os.path.join(ROOT, "ropebwt3_amd", "ropebwt3-amd")
ref = os.path.join(ROOT, "oracle", "_ref", "ropebwt3")
a, ea = (run("amd --host-sort, serial", [amd, "build", "-d", "--host-sort"] + files) if not NOREF else (None, None))
b, eb = run("amd --host-sort -p16", [amd, "build", "-d", "--host-sort", "-p16"] + files)
c2, _ = run("amd --host-sort --rebatch -m40m -p4", [amd, "build", "-d", "--host-sort", "--rebatch", "-m40m", "-p4"] + files)
c3, e3 = run("amd (all on the GPU)", [amd, "build", "-d"] + files)
for l in e3.splitlines():
    if "GPU suffix sorting" in l: print("    " + l)
print("gpu-sort identical:", c3 == b)
if NOREF: a = b
print("amd variants identical:", a == b and a == c2)
if os.path.exists(ref) and not NOREF:
    c, ec = run("reference -t%d" % (os.cpu_count() or 8), [ref, "build", "-d", "-t%d" % min(64, os.cpu_count() or 8)] + files)
    print("IDENTICAL to reference:", a == c)
    # reference merge-only seconds: t("inserted") - t(preceding "constructed partial BWT")
    import re
    tot, last = 0.0, None
    for l in ec.splitlines():
        m = re.match(r"\[M::\w+::([0-9.]+)\*", l)
        if not m: continue
        if "constructed partial BWT" in l: last = float(m.group(1))
        elif "inserted" in l and last is not None: tot += float(m.group(1)) - last; last = None
    print("reference merge-only seconds (sum over rounds): %.2f" % tot)

if SSA:  # sampled suffix array of the index just built: GPU vs the reference's kt_for over strings
    fmd = os.path.join(out, "all.fmd")
    open(fmd, "wb").write(b)
    def runssa(name, cmd):
        t = time.time()
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        print("%-28s %7.2f s  rc=%d  md5=%s  bytes=%d" % (name, time.time() - t, r.returncode, hashlib.md5(r.stdout).hexdigest(), len(r.stdout)), flush=True)
        for l in r.stderr.decode().splitlines():
            if "samples of" in l or "sampled suffix array" in l: print("    " + l)
        return r.stdout
    x = runssa("amd ssa -s8", [amd, "ssa", "-s8", fmd])
    if os.path.exists(ref):
        y = runssa("reference ssa -s8 -t%d" % min(64, os.cpu_count() or 8), [ref, "ssa", "-s8", "-t%d" % min(64, os.cpu_count() or 8), fmd])
        print("SSA IDENTICAL to reference:", x == y)
