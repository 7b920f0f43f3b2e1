#!/usr/bin/env python3
"""profiles/r3_pmc_k_chain_mtb152.json (what bench.py reads as roofline.traffic) from the per-kernel counter table of
tools/pmc_headline.sh:   python tools/pmc_json.py gpurun_out/prof/r3_mtb152_counters.txt profiles/r3_pmc_k_chain_mtb152.json
[kernel name as in the table, default "k_chain<list,mixed,tent>"] [description] [what the bytes are]"""
import json, re, sys
KERNEL = sys.argv[3] if len(sys.argv) > 3 else "k_chain<list,mixed,tent>"
DESC = sys.argv[4] if len(sys.argv) > 4 else "k_chain<list,mixed,tent,text> on the mtb152 headline (bench.py --only headline; %d launches)"
rows = {}
for l in open(sys.argv[1]):
    m = re.match(r"(%s)\s+(\S+)\s+calls\s+(\d+)\s+avg\s+([0-9.]+)" % re.escape(KERNEL), l)
    if m:
        rows[m.group(2)] = (int(m.group(3)), float(m.group(4)))
f, w = rows["FETCH_SIZE"], rows["WRITE_SIZE"]
out = {
    "kernel": DESC % f[0] if "%d" in DESC else DESC,
    "dispatches": f[0], "FETCH_SIZE_KB_per_launch": f[1], "WRITE_SIZE_KB_per_launch": w[1],
    "hbm_bytes_per_launch": int((2.0009 * f[1] + w[1]) * 1024),
    "TCC_HIT_per_launch": rows.get("TCC_HIT_sum", (0, None))[1], "TCC_MISS_per_launch": rows.get("TCC_MISS_sum", (0, None))[1],
    "TCC_EA0_RDREQ_per_launch": rows.get("TCC_EA0_RDREQ_sum", (0, None))[1],
    "correction": "FETCH_SIZE x 2.0009: calibrated on this access pattern (profiles/r4_fetch_size_calibration.json: tools/ubench/lfchase, random 128-byte lines, known bytes; "
                  "the same run gives 128 B per TCC_EA0_RDREQ and x4.0 for random 8-byte stores in WRITE_SIZE, which is reported as is). " + (sys.argv[5] if len(sys.argv) > 5 else
                  "These are bytes behind the L2: an index that fits the 256 MB Infinity Cache is served from there, not from DRAM."),
    "source": "tools/pmc_headline.sh (separate --pmc passes, --kernel-trace only); all counters of the passes: " + sys.argv[1],
}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(out["hbm_bytes_per_launch"])
