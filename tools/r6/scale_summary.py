#!/usr/bin/env python3
"""Summary of a `ropebwt3-amd build -v4` run at scale (tools/r6/scale_*.sh): per merge the engine prints its phases from its HIP events; here they become one
JSON object: per-merge samples as the index passes 1 / 4 / 16 / 64 GB (or the largest reached), totals, Gbp/s of the merge path, k_chain's share of the 208-B roofline,
peak device memory against the index, rows LF-checked.   python tools/r6/scale_summary.py LABEL build.err [wall_seconds]"""
import json
import re
import sys

label, fn = sys.argv[1], sys.argv[2]
wall = float(sys.argv[3]) if len(sys.argv) > 3 else None
err = open(fn, errors="replace").read()
merges = []
for m in re.finditer(r"merge of (\d+) rows into (\d+): (\d+) list slots, (\d+) LF steps.*?fill-to-walkers ([0-9.]+) ms, k_chain ([0-9.]+), settle \+ validation ([0-9.]+), rebuild ([0-9.]+)", err):
    rows, into, slots, steps = (int(m.group(i)) for i in range(1, 5))
    fill, chain, settle, reb = (float(m.group(i)) for i in range(5, 9))
    merges.append({"rows": rows, "into_symbols": into, "lf_steps": steps, "ms_k_chain": chain, "ms_settle_validation": settle, "ms_rebuild": reb, "ms_fill_to_walkers": fill,
                   "gbp_s_merge": round(rows / max(1e-9, (fill + chain + settle + reb) * 1e-3) / 1e9, 2), "k_chain_frac_of_208B_roofline": round(208.0 * rows / max(1e-9, chain * 1e-3) / 8e12, 3)})
if not merges:   # verbose 3: one line per merge from the engine, without the size of the index: it is the running sum
    into = 0
    for m in re.finditer(r"(encoded|merged) the partial BWT for (\d+) symbols|merged (\d+) symbols \((\d+) strings\): lf ([0-9.]+) ms, rank ([0-9.]+) ms \((\d+) LF steps\), rebuild ([0-9.]+) ms", err):
        if m.group(1) == "encoded":
            into += int(m.group(2))
        elif m.group(3):
            rows, lf, rank, steps, reb = int(m.group(3)), float(m.group(5)), float(m.group(6)), int(m.group(7)), float(m.group(8))
            merges.append({"rows": rows, "strings": int(m.group(4)), "into_symbols": into, "lf_steps": steps, "ms_lf": lf, "ms_rank": rank, "ms_rebuild": reb,
                           "gbp_s_merge": round(rows / max(1e-9, (lf + rank + reb) * 1e-3) / 1e9, 2), "rank_frac_of_208B_roofline": round(208.0 * rows / max(1e-9, rank * 1e-3) / 8e12, 3)})
            into += rows
out = {"workload": label, "merges": len(merges)}
tot = re.search(r"GPU merge path: (\d+) symbols merged in ([0-9.]+) ms \(H2D ([0-9.]+) \+ LF ([0-9.]+) \+ rank ([0-9.]+) \+ rebuild ([0-9.]+)\); index ([0-9.]+) MB", err)
if tot:
    out.update(symbols_merged=int(tot.group(1)), merge_path_ms=float(tot.group(2)), ms_h2d=float(tot.group(3)), ms_lf=float(tot.group(4)), ms_rank=float(tot.group(5)), ms_rebuild=float(tot.group(6)),
               index_MB=float(tot.group(7)), merge_path_gbp_s=round(int(tot.group(1)) / float(tot.group(2)) / 1e6, 3))
pk = re.search(r"device memory of the index handle: peak ([0-9.]+) MB, index ([0-9.]+) MB", err)
if pk:
    out.update(handle_peak_MB=float(pk.group(1)), handle_peak_over_index=round(float(pk.group(1)) / max(1.0, float(pk.group(2))), 2))
m = re.search(r"k_chain ([0-9.]+) ms in (\d+) launches, (\d+) steps", err)
if m:
    out.update(k_chain_ms=float(m.group(1)), k_chain_launches=int(m.group(2)), lf_steps=int(m.group(3)),
               k_chain_frac_of_208B_roofline=round(208.0 * out.get("symbols_merged", 0) / (float(m.group(1)) * 1e-3) / 8e12, 3) if out.get("symbols_merged") else None)
m = re.search(r"(\d+) merges redone without tentative records, (\d+) needed the long settle pass; (\d+) rows LF-checked", err)
if m:
    out.update(merges_redone=int(m.group(1)), long_settles=int(m.group(2)), rows_lf_checked=int(m.group(3)))
m = re.search(r"batches: (\d+) \((\d+) symbols\) suffix-sorted on the GPU, (\d+) \((\d+) symbols\) on the host", err)
if m:
    out.update(batches_gpu=int(m.group(1)), symbols_gpu_sorted=int(m.group(2)), batches_host=int(m.group(3)))
m = re.search(r"suffix sorting ([0-9.]+) ms", err)
if m:
    out["gpu_suffix_sorting_ms"] = float(m.group(1))
if wall:
    out["cli_wall_s"] = wall
    if out.get("symbols_merged"):
        out["end_to_end_gbp_s"] = round(out["symbols_merged"] / wall / 1e9, 3)
out["errors"] = [l for l in err.splitlines() if "ERROR" in l or "[E::" in l][:5]
# samples: the last merge below each size mark of the index (symbols -> the engine does not print bytes per merge: marks in symbols)
marks = [1 << 30, 1 << 32, 1 << 33, 1 << 34, 1 << 35]
samples = []
for mk in marks:
    c = [x for x in merges if x["into_symbols"] <= mk]
    if c and (not samples or c[-1] is not samples[-1]):
        samples.append(c[-1])
if merges and (not samples or merges[-1] is not samples[-1]):
    samples.append(merges[-1])
out["samples_as_the_index_grows"] = samples
print(json.dumps(out))
