#!/usr/bin/env python3
"""`ropebwt3-amd build` on K genomes of the synthetic mtb star (tools/gen_mtb.py), default settings and with the given
RB3GPU_* switches; prints wall time, md5 of the .fmd and the CLI's statistics lines.
    python tools/probe_mtb.py K [L] [KEY=VAL ...]      e.g.  python tools/probe_mtb.py 48 4400000 RB3GPU_WINDOW_REBUILD=1"""
import hashlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import gen_mtb
K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
L = int(sys.argv[2]) if len(sys.argv) > 2 and "=" not in sys.argv[2] else 4400000
envs = [a for a in sys.argv[2:] if "=" in a]
out = "/tmp/mtb_star_%d" % L
t = time.time()
files = gen_mtb.generate(K, L, out)
print("generated %d genomes in %.1f s" % (K, time.time() - t), flush=True)
amd = os.path.join(ROOT, "ropebwt3_amd", "ropebwt3-amd")
for env in [{}] + ([dict(e.split("=", 1) for e in envs)] if envs else []):
    for rep in range(2):
        t = time.time()
        r = subprocess.run([amd, "build", "-d"] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
        dt = time.time() - t
        print("%s: %.2f s rc=%d md5=%s bytes=%d" % (env or "default", dt, r.returncode, hashlib.md5(r.stdout).hexdigest(), len(r.stdout)), flush=True)
        nm = 0
        for l in r.stderr.decode().splitlines():
            if "::merge_core" in l:
                nm += 1
                if rep == 1 and (nm % 10 == 0 or nm < 4): print("    round %d: %s" % (nm, l.split("] ", 1)[-1]))
            if "GPU merge path" in l or "GPU suffix sorting" in l or "batches:" in l or "ERROR" in l or "[W" in l or "[E" in l:
                print("    " + l)
