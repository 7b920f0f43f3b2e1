#!/usr/bin/env python3
"""Golden md5 of the reference's .fmd for the synthetic mtb star (tools/gen_mtb.py): runs the unmodified reference binary
(oracle/_ref/ropebwt3, built from /root/reference by oracle/Makefile) on the first K genomes, one file per batch as the
reference is meant to be run, and records md5, size and the reference's own timing in tests/golden/MANIFEST.json under
"mtb_star".  K = 152 takes ~14 minutes on 8 cores (rb3_fmi_merge_plain has two chains of work per round).
    python tools/make_golden_mtb.py 24 152"""
import hashlib, json, os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import gen_mtb
L = 4400000
ref = os.path.join(ROOT, "oracle", "_ref", "ropebwt3")
out = "/tmp/mtb_star_%d" % L
man_fn = os.path.join(ROOT, "tests", "golden", "MANIFEST.json")
man = json.load(open(man_fn))
ent = man.setdefault("mtb_star", {"generator": "tools/gen_mtb.py", "genome_len": L, "flags": ["-d"], "prefixes": {}})
for K in [int(a) for a in sys.argv[1:]]:
    files = gen_mtb.generate(K, L, out)
    t = time.time()
    r = subprocess.run([ref, "build", "-d", "-t%d" % (os.cpu_count() or 8)] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    dt = time.time() - t
    tot, last = 0.0, None
    for l in r.stderr.decode().splitlines():
        m = re.match(r"\[M::\w+::([0-9.]+)\*", l)
        if not m:
            continue
        if "constructed partial BWT" in l:
            last = float(m.group(1))
        elif "inserted" in l and last is not None:
            tot += float(m.group(1)) - last
            last = None
    ent["prefixes"][str(K)] = {"fmd_md5": hashlib.md5(r.stdout).hexdigest(), "fmd_bytes": len(r.stdout),
                               "reference_seconds": round(dt, 1), "reference_merge_only_seconds": round(tot, 1), "reference_threads": os.cpu_count() or 8,
                               "n_symbols": None}
    print(K, ent["prefixes"][str(K)], flush=True)
    json.dump(man, open(man_fn, "w"), indent=1, sort_keys=True)
