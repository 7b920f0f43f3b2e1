#!/usr/bin/env python3
"""Golden md5 of the reference's .fmd for an index of MORE THAN 2^32 symbols: the first N haplotypes of
tools/gen_family.py's big family (180 Mbp each, contigs of 20-100 Mbp, both strands: 360 M symbols per haplotype; N = 12:
4.32 G symbols), built by the unmodified reference binary oracle/_ref/ropebwt3.  Recorded in tests/golden/MANIFEST.json under
"big_index".     python tools/make_golden_big.py [N [L]]"""
import hashlib, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import gen_family
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
L = int(sys.argv[2]) if len(sys.argv) > 2 else 180000000
ref = os.path.join(ROOT, "oracle", "_ref", "ropebwt3")
files = gen_family.big_haplotype_files(N, L, "/tmp/big_family_%d" % L)
t = time.time()
p = subprocess.Popen([ref, "build", "-d", "-t8"] + files, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
md5, nb = hashlib.md5(), 0
while True:
    b = p.stdout.read(1 << 24)
    if not b:
        break
    md5.update(b), 
    nb += len(b)
assert p.wait() == 0
man_fn = os.path.join(ROOT, "tests", "golden", "MANIFEST.json")
man = json.load(open(man_fn))
ent = man.setdefault("big_index", {"generator": "tools/gen_family.py big_haplotype_files"})
ent["%dx%d" % (N, L)] = {"haplotypes": N, "genome_len": L, "symbols": 2 * N * L + 2 * sum(1 for fn in files for l in open(fn, "rb") if l[:1] == b">"),
                         "fmd_md5": md5.hexdigest(), "fmd_bytes": nb, "reference_seconds": round(time.time() - t, 1), "reference_threads": 8,
                         "note": "oracle/_ref/ropebwt3 build -d -t8 (one batch per file)"}
print(ent["%dx%d" % (N, L)], flush=True)
json.dump(man, open(man_fn, "w"), indent=1, sort_keys=True)
