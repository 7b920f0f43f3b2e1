#!/usr/bin/env python3
"""Static instruction mix of every kernel in a gfx950 assembly file (hipcc --cuda-device-only -S).
usage: isa_stats.py file.s [substring-of-demangled-name]"""
import re, sys, subprocess, collections
path = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ""
lines = open(path).read().split("\n")
kern = None; stats = {}; order = []
meta = {}
for ln in lines:
    m = re.match(r"^(_Z\w+):\s*(;.*)?$", ln)
    if m and not ln.startswith("."):
        kern = m.group(1); stats[kern] = collections.Counter(); order.append(kern); continue
    if kern is None: continue
    s = ln.strip()
    if s.startswith(".amdhsa_next_free_vgpr"): meta.setdefault(kern, {})["vgpr"] = s.split()[-1]
    if s.startswith(".amdhsa_group_segment_fixed_size"): meta.setdefault(kern, {})["lds"] = s.split()[-1]
    if s.startswith(".amdhsa_accum_offset"): meta.setdefault(kern, {})["acc"] = s.split()[-1]
    if s.startswith(".end_amdhsa_kernel") : kern = None; continue
    if not s or s.startswith((";", ".", "//")) or s.endswith(":"): continue
    op = s.split()[0]
    c = stats[kern]
    if op.startswith("v_"): c["valu"] += 1; c["dpp"] += ("dpp" in s or "row_" in s or "quad_perm" in s)
    elif op.startswith("s_"):
        if op.startswith(("s_cbranch", "s_branch")): c["branch"] += 1
        elif op.startswith("s_waitcnt"): c["wait"] += 1
        elif op.startswith("s_nop"): c["nop"] += 1
        else: c["salu"] += 1
    elif op.startswith("ds_"): c["lds"] += 1
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): c["vmem"] += 1
    else: c["other"] += 1
names = subprocess.run(["c++filt"], input="\n".join(order), capture_output=True, text=True).stdout.split("\n")
print("%-70s %6s %6s %5s %5s %5s %5s %5s %5s %6s" % ("kernel", "valu", "salu", "lds", "vmem", "br", "wait", "nop", "vgpr", "ldsB"))
for k, n in zip(order, names):
    if k not in meta or pat not in n: continue
    c = stats[k]
    print("%-70s %6d %6d %5d %5d %5d %5d %5d %5s %6s" % (n[:70], c["valu"], c["salu"], c["lds"], c["vmem"], c["branch"], c["wait"], c["nop"], meta[k].get("vgpr"), meta[k].get("lds")))
