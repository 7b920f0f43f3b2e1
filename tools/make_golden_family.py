#!/usr/bin/env python3
"""Golden md5 of the reference's .fmd for the inputs of tools/gen_family.py (unmodified reference binary oracle/_ref/ropebwt3,
built from /root/reference by oracle/Makefile); recorded in tests/golden/MANIFEST.json under "family".
    python tools/make_golden_family.py"""
import hashlib, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import gen_family
ref = os.path.join(ROOT, "oracle", "_ref", "ropebwt3")
man_fn = os.path.join(ROOT, "tests", "golden", "MANIFEST.json")
man = json.load(open(man_fn))
ent = man.setdefault("family", {"generator": "tools/gen_family.py"})
CASES = {"haplotypes_4x25M": ("haplotypes", 4, 25000000, 6000000), "relatives_320x200k": ("relatives", 320, 200000)}
for name, spec in CASES.items():
    if len(sys.argv) > 1 and name not in sys.argv[1:]:
        continue
    if spec[0] == "haplotypes":
        files = gen_family.haplotypes(spec[1], spec[2], spec[3], "/tmp/family_hap")
    else:
        files = [gen_family.relatives(spec[1], spec[2], "/tmp/family_rel.fa")]
    t = time.time()
    r = subprocess.run([ref, "build", "-d", "-t8"] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-500:]
    ent[name] = {"spec": list(spec), "fmd_md5": hashlib.md5(r.stdout).hexdigest(), "fmd_bytes": len(r.stdout), "reference_seconds": round(time.time() - t, 1), "reference_threads": 8,
                 "note": "oracle/_ref/ropebwt3 build -d -t8 (one batch per file at the default -m7g; the .fmd does not depend on the batching)"}
    print(name, ent[name], flush=True)
    json.dump(man, open(man_fn, "w"), indent=1, sort_keys=True)
