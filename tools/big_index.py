#!/usr/bin/env python3
"""An index of MORE THAN 2^32 symbols on one MI355X (BASELINE configs[3]/[4] live in this regime): N haplotypes
of a 180 Mbp genome (0.1 % substitutions each, contigs of 20-100 Mbp, both strands: 360 M symbols per batch; 24 haplotypes =
8.64 G symbols), one haplotype per merge round, in process through the C ABI -- the GPU sorter's text-order words + suffix array,
rb3gpu_merge_text_sa_dev, as the CLI runs it.  Beyond 2^32 symbols the slot headers count from the group start (the walk reads
the 64-byte directory entry again: three lines per rank instead of two) and the common step of k_chain runs on 64-bit positions.
Then a batch of reads into the finished index (config 4's shape: one walker per read, records in text order).

    python tools/big_index.py [--hap 24] [--len 180000000] [--reads 1000000] [--lf-check N] [--fmd-md5]

Prints one JSON object: per-round times of the last rounds, Gbp/s, k_chain's roofline, the index's bytes and the handle's peak
device memory.  bench.py runs it as the leg `aux_index_8g`; tests/test_gpu_engine.py::test_index_beyond_2_32_symbols uses
build() with the every-row LF check and compares the .fmd of 12 haplotypes (4.32 G symbols) with the reference's md5."""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

_LUT = np.full(256, 5, dtype=np.uint8)
for _i, _c in enumerate(b"ACGT"):
    _LUT[_c] = _i + 1
_COMP = np.array([0, 4, 3, 2, 1, 5], dtype=np.uint8)


def batch_text(contigs):
    """io.c:84-102 for a list of ASCII contigs: nt6(forward), 0, nt6(reverse complement), 0 per record; returns (text, strings)"""
    n = sum(2 * (c.size + 1) for c in contigs)
    t = np.empty(n, dtype=np.uint8)
    p = 0
    for c in contigs:
        f = _LUT[c]
        t[p:p + f.size] = f
        t[p + f.size] = 0
        p += f.size + 1
        t[p:p + f.size] = _COMP[f[::-1]]
        t[p + f.size] = 0
        p += f.size + 1
    return t, 2 * len(contigs)


def build(n_hap, L, lf_check=None, log=None, device=0):
    """builds the index; returns (handle, sorter, per-round records, base genome)"""
    from ropebwt3_amd import Rb3Gpu, Sorter, host, walker_step
    from tools import gen_family
    h = Rb3Gpu(device=device, verbose=0)
    if lf_check is not None:
        h.tune("lf_check", lf_check)
    srt = Sorter(device)
    base = gen_family.big_base(L)
    rounds = []
    prev = h.stats()
    for k in range(n_hap):
        t0 = time.time()
        text, n_seq = batch_text(gen_family.big_contigs(base, k))
        t1 = time.time()
        srt.upload(text)
        d, dtw, dsa = srt.sort_uploaded_sa(text.size)
        t2 = time.time()
        if k == 0:
            h.from_plain_dev(d, text.size)
        else:
            h.merge_text_dev(d, dtw, text.size, host.walkers_text(text, walker_step(device, text.size, n_seq)), commit=True, d_sa=dsa)
        t3 = time.time()
        srt.release(d)
        s = h.stats()
        r = {"round": k, "symbols": int(text.size), "strings": n_seq, "index_symbols": int(h.get_acc()[6]), "gen_s": round(t1 - t0, 2), "sort_s": round(t2 - t1, 3), "merge_s": round(t3 - t2, 3),
             "ms_lf": round(s["ms_lf"] - prev["ms_lf"], 3), "ms_rank": round(s["ms_rank"] - prev["ms_rank"], 3), "ms_chain": round(s["ms_chain"] - prev["ms_chain"], 3),
             "ms_rebuild": round(s["ms_build"] - prev["ms_build"], 3), "lf_steps": int(s["n_lf_steps"] - prev["n_lf_steps"]), "fallbacks": int(s["n_fallbacks"] - prev["n_fallbacks"]),
             "index_bytes": int(s["bytes_index"]), "peak_bytes": int(s["bytes_peak"]), "rebuild_emitted_again": int(s["n_reb_again"] - prev["n_reb_again"]), "ms_alloc": round(s["ms_alloc"] - prev["ms_alloc"], 1)}
        if k == n_hap - 1:   # what the handle holds after its last round, buffer by buffer
            r["buffers_MB"] = {n: round(b / 1e6, 1) for n, b in sorted(h.buffers().items(), key=lambda x: -x[1])}
        rounds.append(r)
        prev = s
        if log:
            log("big index: haplotype %d/%d merged: %d symbols in the index (%.0f MB), rank %.1f ms (k_chain %.1f), rebuild %.1f ms%s, sort %.2f s, allocations %.0f ms" %
                (k + 1, n_hap, r["index_symbols"], r["index_bytes"] / 1e6, r["ms_rank"], r["ms_chain"], r["ms_rebuild"], " (emitted twice)" if r["rebuild_emitted_again"] else "", r["sort_s"], r["ms_alloc"]))
    return h, srt, rounds, base


def reads_into(h, base, n_reads, reps=3, seed=31):
    """a batch of 150 bp reads (1 % errors) drawn from the base genome, merged (not committed) into the index"""
    rng = np.random.default_rng(seed)
    g = _LUT[base]
    st = rng.integers(0, g.size - 150, size=n_reads)
    r = np.stack([g[s:s + 150] for s in st])
    m = rng.random(r.shape) < 0.01
    r[m] = rng.integers(1, 5, size=int(m.sum()), dtype=np.uint8)
    from tests import util
    t2 = util.make_text(list(r))
    d, d_tw, d_sa = h.sort_text_sa(t2)
    h.merge_text_dev(d, d_tw, t2.size, 2 * n_reads, commit=False, d_sa=d_sa)
    h.stats_reset()
    t = time.perf_counter()
    for _ in range(reps):
        h.merge_text_dev(d, d_tw, t2.size, 2 * n_reads, commit=False, d_sa=d_sa)
    dt = (time.perf_counter() - t) / reps
    s = h.stats()
    for p in (d, d_tw, d_sa):
        h.dev_free(p)
    return {"symbols": int(t2.size), "strings": 2 * n_reads, "ms_per_merge": round(dt * 1e3, 3), "Gbp/s": round(t2.size / dt / 1e9, 3),
            "ms_lf": round(s["ms_lf"] / reps, 3), "ms_rank": round(s["ms_rank"] / reps, 3), "ms_chain": round(s["ms_chain"] / reps, 3), "ms_rebuild": round(s["ms_build"] / reps, 3),
            "lf_steps": int(s["n_lf_steps"] // reps), "fallbacks": int(s["n_fallbacks"]), "peak_bytes": int(s["bytes_peak"])}


def fmd_md5(h):
    """the .fmd the CLI would write for this index (data section packed on the device, rank index on the host)"""
    from ropebwt3_amd import host
    data = host.fmd_bytes_from_words(h.export_fmd_words(), h.get_acc())
    return hashlib.md5(data).hexdigest()


def summary(rounds, reads, n_hap, L, t_total):
    """the leg of bench.py: the last rounds of the build (index > 2^32 symbols) and the reads batch, each with k_chain's roofline"""
    ALGO = 208
    last = [r for r in rounds if r["round"] > 0 and r["index_symbols"] - r["symbols"] >= (1 << 32)] or rounds[-1:]
    sym = sum(r["symbols"] for r in last)
    ms = sum(r["ms_lf"] + r["ms_rank"] + r["ms_rebuild"] for r in last)
    steps, msc = sum(r["lf_steps"] for r in last), sum(r["ms_chain"] for r in last)
    out = {"workload": "%d haplotypes x %d bp (0.1 %% substitutions, contigs of 20-100 Mbp, both strands), one haplotype = 360 M symbols per merge round, into an index that grows to %d symbols "
                       "(> 2^32: slot headers relative to the group, 64-bit walker step); then %d x 150 bp reads into it" % (n_hap, L, rounds[-1]["index_symbols"], reads["strings"] // 2 if reads else 0),
           "rounds_beyond_2^32": len(last), "Gbp/s_merge_path_beyond_2^32": round(sym / (ms * 1e-3) / 1e9, 3) if ms > 0 else None,
           "ms_per_round_beyond_2^32": {k: round(sum(r[k] for r in last) / len(last), 3) for k in ("ms_lf", "ms_rank", "ms_chain", "ms_rebuild")},
           "roofline": {"bound": "hbm", "kernel": "k_chain (long strings, run-coded index beyond 2^32 symbols)", "achieved": round(ALGO * steps / (msc * 1e-3) / 1e9, 1) if msc > 0 else None, "peak": 8000.0, "unit": "GB/s",
                        "frac": round(ALGO * steps / (msc * 1e-3) / 1e9 / 8000.0, 4) if msc > 0 else None, "lf_steps_per_s": round(steps / (msc * 1e-3) / 1e9, 3) if msc > 0 else None,
                        "ms_per_launch": round(msc / len(last), 3)},
           "residency": {"index_symbols": rounds[-1]["index_symbols"], "index_bytes": rounds[-1]["index_bytes"], "bytes_per_symbol": round(rounds[-1]["index_bytes"] / rounds[-1]["index_symbols"], 4),
                         "peak_device_bytes_of_the_handle": max(r["peak_bytes"] for r in rounds), "sorter_scratch": "52 B per batch symbol beside it (its own object)",
                         "handle_peak_bytes_per_batch_symbol": round((max(r["peak_bytes"] for r in rounds) - rounds[-1]["index_bytes"]) / max(1, max(r["symbols"] for r in rounds)), 1),
                         "buffers_MB_after_the_last_round": rounds[-1].get("buffers_MB")},
           "fallbacks": sum(r["fallbacks"] for r in rounds), "wall_s": round(t_total, 1),
           "first_and_last_rounds": rounds[:2] + rounds[-2:]}
    if reads:
        out["reads_into_it"] = dict(reads, roofline={"bound": "hbm", "kernel": "k_chain + k_pos_finalize_check_rows (the rank phase; records in text order)",
                                                     "achieved": round(ALGO * reads["lf_steps"] / (reads["ms_rank"] * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                                                     "frac": round(ALGO * reads["lf_steps"] / (reads["ms_rank"] * 1e-3) / 1e9 / 8000.0, 4), "k_chain_ms_per_launch": reads["ms_chain"]})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hap", type=int, default=24)
    ap.add_argument("--len", type=int, default=180000000)
    ap.add_argument("--reads", type=int, default=1000000)
    ap.add_argument("--lf-check", type=int, default=None)
    ap.add_argument("--fmd-md5", action="store_true")
    a = ap.parse_args()
    t0 = time.time()
    h, srt, rounds, base = build(a.hap, a.len, a.lf_check, log=lambda m: print("[big] " + m, file=sys.stderr, flush=True))
    rd = reads_into(h, base, a.reads) if a.reads > 0 else None
    out = summary(rounds, rd, a.hap, a.len, time.time() - t0)
    if a.fmd_md5:
        out["fmd_md5"] = fmd_md5(h)
    srt.close()
    h.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
