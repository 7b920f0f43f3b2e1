#!/usr/bin/env python3
"""bench.py -- merge-path throughput of the MI355X engine (BASELINE.json metric:
"Gbp/sec indexed (build merge)").

A STEP is one pass of the hot path over one batch: LF array of the partial BWT B2 + all LF walkers
against the accumulated BWT B1 (rank) + interleave/rebuild of the block array, with B1 and B2
already resident in HBM and the result discarded (commit=0) so that every step does identical work.

N=1 workload = BASELINE.json configs[1] as SURVEY 8(d) defines it without network access:
G0 = 4.4 Mbp of uniform random ACGT (seed 1), G1 = G0 with 0.1 % substitutions (seed 2); the step
merges G1 (both strands, 8,800,002 symbols, 2 strings) into the index of G0.

`value` is measured through the entry point that takes exactly what the reference's
rb3_fmi_merge_plain(r, len, seq, n_threads) takes -- the partial BWT and nothing else
(rb3gpu_merge_plain_dev; fm-index.c:279).  The same JSON line also carries
  * aux_entry_points: the same step with the host buffer and the PCIe copy inside (rb3gpu_merge_plain), and through the
    entry points that take the inverse suffix array of the batch next to its BWT (what the CLI uses: its batches are
    suffix-sorted on the GPU, so the inverse suffix array is in HBM anyway);
  * target_workload: BASELINE configs[2] (mtb152: 152 genomes of 4.4 Mbp, synthetic star of tools/gen_mtb.py) end to end
    through `ropebwt3-amd build`, one file per batch as the reference is run, .fmd md5 checked against the reference's
    (tests/golden/MANIFEST.json), with the merge-path time, the whole-build time and per-kernel rooflines, next to the
    unmodified reference timed here on a stated prefix of the same files;
  * aux_reads_regime / aux_large_index: the chain kernel where it is bound by memory rather than latency, the second on
    an index far larger than L2 + Infinity Cache.

N>1 (weak scaling, one process per GPU): see ropebwt3_amd/multi.py -- interval-sharded index (north_star) or partitioned
input + tree merge; value = symbols merged by all ranks / max-over-ranks time of the whole sharded step.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_STEP = 208          # 16 B row entry r/w + 64 B directory line + 128 B block line per LF step
ENGINE_BYTES_PER_STEP = {"plain": 152, "rows": 152, "text": 144}   # what this engine moves per step: 128-B slot + 8-B record (+ 16-B row word r/w | 8-B text word)
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench] " + msg, file=sys.stderr, flush=True)


def gen_genomes(n, rate, seed0, seeds):
    from tests import util
    g0 = util.random_genome(np.random.default_rng(seed0), n)
    return g0, [util.mutate(np.random.default_rng(s), g0, rate) for s in seeds]


def cpu_baseline(b1, b2):
    """Time the merge of the SAME step on the host cores: the unmodified reference
    (oracle/_ref/librb3ref.so: rb3_enc_plain2fmr + rb3_fmi_merge_plain) when it travelled with
    the repository, else the OpenMP port in oracle/liboracle.so.  Checker/baseline only."""
    from tests import util
    cores = os.cpu_count() or 1
    try:
        ref = util.Reference()
        r = ref.L.rb3_enc_plain2fmr(b1.size, b1.ctypes.data, 0, 0, cores)
        t = time.time()
        ref.L.rb3_fmi_merge_plain(r, b2.size, b2.ctypes.data, cores)
        dt = time.time() - t
        ref.L.mr_destroy(r)
        kind = "reference"
    except (FileNotFoundError, OSError):
        orc = util.Oracle()
        t = time.time()
        orc.mg_rank(b1, b2, cores)
        dt = time.time() - t
        kind = "port"
    n_str = int((b2 == 0).sum())
    return {"value": b2.size / dt / 1e9, "unit": "Gbp/s", "cores": cores, "kind": kind, "seconds": round(dt, 3),
            "sample": "the full N=1 step (%d symbols, %d strings -> only %d of the %d threads offered have work, as in the reference's kt_for over strings)" % (b2.size, n_str, min(n_str, cores), cores)}


def load_pmc_traffic(kernel):
    """HBM bytes per launch of a kernel from the committed rocprofv3 --pmc passes (profiles/), if any."""
    for fn in ("r2_pmc_%s.json" % kernel, "r1_pmc_%s.json" % kernel):
        try:
            return json.load(open(os.path.join(ROOT, "profiles", fn))).get("hbm_bytes_per_launch")
        except (OSError, ValueError):
            pass
    return None


def chain_roofline(rows_per_launch, ms_chain, mode, traffic, note):
    algo = ALGO_BYTES_PER_STEP * rows_per_launch
    ach = algo / (ms_chain * 1e-3) / 1e9
    eng = ENGINE_BYTES_PER_STEP[mode] * rows_per_launch / (ms_chain * 1e-3) / 1e9
    d = {"bound": "hbm", "kernel": "k_chain", "achieved": round(ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 6),
         "traffic": traffic, "algorithmic_bytes_per_launch": algo, "ms_per_launch": round(ms_chain, 4),
         "lf_steps_per_s": round(rows_per_launch / (ms_chain * 1e-3) / 1e9, 3),
         "achieved_engine": round(eng, 3), "frac_engine": round(eng / HBM_PEAK_GBS, 6),
         "achieved_counter": round(traffic / (ms_chain * 1e-3) / 1e9, 3) if traffic else None,
         "frac_counter": round(traffic / (ms_chain * 1e-3) / 1e9 / HBM_PEAK_GBS, 6) if traffic else None,
         "note": note}
    return d


def reads_regime(h_factory, n_reads, seed=11):
    """Auxiliary measurement in the many-short-strings regime (where the chain kernel is bound by HBM
    bandwidth rather than latency): merge n_reads x 150 bp reads (both strands) into an index of as many."""
    from ropebwt3_amd import host
    from tests import util
    rng = np.random.default_rng(seed)
    g = util.random_genome(rng, 10 * n_reads)
    st = rng.integers(0, len(g) - 150, size=2 * n_reads)

    def reads(idx):
        r = np.stack([g[s:s + 150] for s in idx])
        m = rng.random(r.shape) < 0.01
        r[m] = rng.integers(1, 5, size=int(m.sum()), dtype=np.uint8)
        return list(r)
    b1 = host.build_bwt(util.make_text(reads(st[:n_reads])))
    t2 = util.make_text(reads(st[n_reads:]))
    h = h_factory()
    h.from_plain(b1)
    d, d_tw = h.sort_text(t2)   # BWT + text-order words; one walker per string, made on the device
    b2 = t2
    h.merge_text_dev(d, d_tw, b2.size, 2 * n_reads, commit=False)
    h.stats_reset()
    reps = 5
    t = time.perf_counter()
    for _ in range(reps):
        h.merge_text_dev(d, d_tw, b2.size, 2 * n_reads, commit=False)
    dt = (time.perf_counter() - t) / reps
    st_ = h.stats()
    ms_chain = st_["ms_chain"] / reps
    h.dev_free(d)
    h.dev_free(d_tw)
    h.close()
    return {"workload": "reads regime: merge %d x 150 bp reads (both strands, %d symbols, %d strings) into an index of %d symbols" % (n_reads, b2.size, 2 * n_reads, b1.size),
            "value": round(b2.size / dt / 1e9, 4), "unit": "Gbp/s", "ms_per_step": round(dt * 1e3, 3),
            "roofline": chain_roofline(b2.size, ms_chain, "text", None, "index of %.0f MB: resident in L2 + Infinity Cache" % (st_["bytes_index"] / 1e6))}


def large_index_regime(h_factory, n_index, n_reads, seed=21):
    """The chain kernel against an index that is far larger than L2 + Infinity Cache: a random genome of
    n_index / 2 bp (both strands: n_index symbols, 0.5 B per symbol of bit-plane slots), suffix-sorted on the GPU, and one
    batch of reads drawn from it (one walker per read).  Every rank then hits a slot that comes from HBM."""
    from tests import util
    rng = np.random.default_rng(seed)
    g = util.random_genome(rng, n_index // 2 - 1)
    h = h_factory()
    t0 = time.time()
    d, d_tw = h.sort_text(util.make_text([g]))
    h.dev_free(d_tw)
    h.from_plain_dev(d, n_index)
    h.dev_free(d)
    t_idx = time.time() - t0
    st = rng.integers(0, len(g) - 150, size=n_reads)
    r = np.stack([g[s:s + 150] for s in st])
    m = rng.random(r.shape) < 0.01
    r[m] = rng.integers(1, 5, size=int(m.sum()), dtype=np.uint8)
    t2 = util.make_text(list(r))
    d, d_tw = h.sort_text(t2)
    h.merge_text_dev(d, d_tw, t2.size, 2 * n_reads, commit=False)
    h.stats_reset()
    reps = 3
    t = time.perf_counter()
    for _ in range(reps):
        h.merge_text_dev(d, d_tw, t2.size, 2 * n_reads, commit=False)
    dt = (time.perf_counter() - t) / reps
    st_ = h.stats()
    ms_chain = st_["ms_chain"] / reps
    out = {"workload": "large index: merge %d x 150 bp reads (both strands, %d symbols) into the index of a random genome, %d symbols = %.0f MB of slots in HBM (index built in %.1f s incl. GPU suffix sorting)" % (n_reads, t2.size, n_index, st_["bytes_index"] / 1e6, t_idx),
           "value": round(t2.size / dt / 1e9, 4), "unit": "Gbp/s", "ms_per_step": round(dt * 1e3, 3),
           "phases_ms_per_step": {"lf": round(st_["ms_lf"] / reps, 3), "rank": round(st_["ms_rank"] / reps, 3), "rebuild": round(st_["ms_build"] / reps, 3)},
           "rebuild_streaming": {"bytes_per_step": int(st_["bytes_rebuild"] // reps), "GB/s": round(st_["bytes_rebuild"] / max(1e-9, st_["ms_build"]) / 1e6, 1),
                                 "frac": round(st_["bytes_rebuild"] / max(1e-9, st_["ms_build"]) / 1e6 / HBM_PEAK_GBS, 4)},
           "roofline": chain_roofline(t2.size, ms_chain, "text", load_pmc_traffic("k_chain_large"), "every slot read comes from HBM (index >> 256 MB of L2 + Infinity Cache)")}
    h.dev_free(d)
    h.dev_free(d_tw)
    h.close()
    return out


def parse_cli_stats(err):
    """the statistics lines `ropebwt3-amd build` prints at verbosity 3"""
    d = {}
    m = re.search(r"GPU merge path: (\d+) symbols merged in ([0-9.]+) ms \(H2D ([0-9.]+) \+ LF ([0-9.]+) \+ rank ([0-9.]+) \+ rebuild ([0-9.]+)\); index ([0-9.]+) MB", err)
    if m:
        d.update(symbols_merged=int(m.group(1)), merge_path_ms=float(m.group(2)), h2d_ms=float(m.group(3)), lf_ms=float(m.group(4)), rank_ms=float(m.group(5)),
                 rebuild_ms=float(m.group(6)), index_mb=float(m.group(7)))
    m = re.search(r"GPU sorter threads: text upload ([0-9.]+) ms, suffix sorting ([0-9.]+) ms", err)
    if m:
        d.update(text_upload_ms=float(m.group(1)), sort_ms=float(m.group(2)))
    m = re.search(r"rebuild: ([0-9.]+) ms for (\d+) algorithmic bytes.*k_chain ([0-9.]+) ms in (\d+) launches, (\d+) steps", err)
    if m:
        d.update(rebuild_algo_bytes=int(m.group(2)), chain_ms=float(m.group(3)), chain_launches=int(m.group(4)), lf_steps=int(m.group(5)))
    m = re.search(r"batches: (\d+) \((\d+) symbols\) suffix-sorted on the GPU, (\d+) \(", err)
    if m:
        d.update(batches_gpu=int(m.group(1)), batches_host=int(m.group(3)))
    m = re.search(r"Real time: ([0-9.]+) sec", err)
    if m:
        d["cli_real_time_s"] = float(m.group(1))
    return d


def target_workload(K, L, ref_prefix, keep=None):
    """BASELINE configs[2] end to end through the CLI."""
    from tools import gen_mtb
    from ropebwt3_amd import _build
    man = json.load(open(os.path.join(ROOT, "tests", "golden", "MANIFEST.json"))).get("mtb_star", {})
    gold = man.get("prefixes", {}).get(str(K)) if L == man.get("genome_len") else None
    tmp = keep or tempfile.mkdtemp(prefix="rb3_mtb_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    t = time.time()
    files = gen_mtb.generate(K, L, tmp)
    t_gen = time.time() - t
    best = None
    for rep in range(2):   # the first run pays for the page cache and the HIP start-up; report the second
        t = time.time()
        r = subprocess.run([_build.BIN_CLI, "build", "-d"] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        wall = time.time() - t
        if r.returncode != 0:
            return {"error": r.stderr.decode()[-400:]}
        best = (wall, r)
    wall, r = best
    md5 = hashlib.md5(r.stdout).hexdigest()
    st = parse_cli_stats(r.stderr.decode())
    nsym = st.get("symbols_merged", 0)
    mp = st.get("merge_path_ms", 0.0)
    up = st.get("text_upload_ms", 0.0)
    out = {"workload": "cfg3-synthetic-mtb%d: `ropebwt3-amd build -d g000.fa ... g%03d.fa`, %d genomes of %d bp (star phylogeny, 0.1 %% substitutions + 10 indels each; tools/gen_mtb.py), one file per batch = %d merge rounds" % (K, K - 1, K, L, K - 1),
           "symbols_merged": nsym, "fmd_bytes": len(r.stdout), "fmd_md5": md5,
           "fmd_identical_to_reference": (md5 == gold["fmd_md5"]) if gold else None,
           "reference_fmd_md5_source": "tests/golden/MANIFEST.json mtb_star/%d (oracle/_ref/ropebwt3, tools/make_golden_mtb.py)" % K if gold else "no golden for this size",
           "build_wall_s": round(wall, 3), "generate_s": round(t_gen, 2),
           "merge_path": {"ms": round(mp, 3), "Gbp/s": round(nsym / mp / 1e6, 4) if mp else None,
                          "ms_incl_text_upload": round(mp + up, 3), "Gbp/s_incl_text_upload": round(nsym / (mp + up) / 1e6, 4) if mp else None,
                          "phases_ms": {"text_upload(H2D)": up, "lf": st.get("lf_ms"), "rank": st.get("rank_ms"), "rebuild": st.get("rebuild_ms")},
                          "definition": "SURVEY 8(d): H2D + rank + interleave + rebuild summed over the rounds, suffix sorting and file I/O excluded; with GPU suffix sorting the H2D of a batch is its text upload (the BWT never crosses PCIe)"},
           "suffix_sorting_ms_overlapped": st.get("sort_ms"), "index_mb": st.get("index_mb"), "batches_sorted_on_gpu": st.get("batches_gpu"), "batches_sorted_on_host": st.get("batches_host")}
    # (b) of SURVEY 8(d) config 3: the same files re-batched (a batch spans input files: ~16 rounds instead of 151); same .fmd
    t = time.time()
    rb = subprocess.run([_build.BIN_CLI, "build", "-d", "--rebatch", "-m80m"] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    wall_b = time.time() - t
    if rb.returncode == 0:
        sb = parse_cli_stats(rb.stderr.decode())
        out["rebatched"] = {"command": "ropebwt3-amd build -d --rebatch -m80m (9-10 genomes per batch)", "build_wall_s": round(wall_b, 3), "merge_rounds": sb.get("chain_launches"),
                            "merge_path_ms": sb.get("merge_path_ms"), "phases_ms": {"text_upload(H2D)": sb.get("text_upload_ms"), "lf": sb.get("lf_ms"), "rank": sb.get("rank_ms"), "rebuild": sb.get("rebuild_ms")},
                            "Gbp/s_merge_path": round(sb.get("symbols_merged", 0) / sb["merge_path_ms"] / 1e6, 4) if sb.get("merge_path_ms") else None,
                            "fmd_identical": hashlib.md5(rb.stdout).hexdigest() == md5}
    if st.get("chain_ms") and st.get("chain_launches"):
        rows = nsym / max(1, st["chain_launches"])
        ms = st["chain_ms"] / st["chain_launches"]
        out["roofline_k_chain_mixed"] = chain_roofline(int(rows), ms, "text", None, "k_chain<list,mixed,tent,text>: average over the %d merge rounds (run-coded index, intervals of up to 152 matching suffixes); VALU-bound at ~250 vector instructions per 8-walker step (profiles/r2_sq_mtb*.txt)" % st["chain_launches"])
    if st.get("rebuild_algo_bytes") and st.get("rebuild_ms"):
        gbs = st["rebuild_algo_bytes"] / st["rebuild_ms"] / 1e6
        out["roofline_rebuild"] = {"bound": "hbm", "kernel": "k_reb_group + k_place (+ window kernels on the groups they leave)", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 5),
                                   "algorithmic_bytes": st["rebuild_algo_bytes"], "ms": st["rebuild_ms"], "note": "streaming roofline: 9 B per batch row + old block array + new block array per round, summed over the rounds"}
    # the unmodified reference on a prefix of the same files (the whole set takes 826 s on 8 cores: recorded in the manifest)
    ref = os.path.join(ROOT, "oracle", "_ref", "ropebwt3")
    if ref_prefix > 1 and os.path.exists(ref):
        cores = os.cpu_count() or 1
        t = time.time()
        rr = subprocess.run([ref, "build", "-d", "-t%d" % min(cores, 64)] + files[:ref_prefix], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        dt = time.time() - t
        tot, last, nsr = 0.0, None, 0
        for l in rr.stderr.decode().splitlines():
            m = re.match(r"\[M::\w+::([0-9.]+)\*", l)
            if not m:
                continue
            if "constructed partial BWT" in l:
                last = float(m.group(1))
            elif "inserted" in l and last is not None:
                tot += float(m.group(1)) - last
                last = None
                nsr += int(re.search(r"inserted (\d+) symbols", l).group(1))
        ra = subprocess.run([_build.BIN_CLI, "build", "-d"] + files[:ref_prefix], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        sa = parse_cli_stats(ra.stderr.decode())
        out["cpu_baseline"] = {"value": round(nsr / tot / 1e9, 6) if tot > 0 else None, "unit": "Gbp/s", "cores": cores, "kind": "reference",
                               "sample": "`ropebwt3 build -d -t%d` (oracle/_ref, unmodified) on the first %d of the %d files: merge-only seconds (its own timers: 'inserted' minus 'constructed partial BWT', summed over %d rounds) %.2f s of %.1f s; rb3_fmi_merge_plain has one chain per string, i.e. 2 threads of work per round" % (min(cores, 64), ref_prefix, K, ref_prefix - 1, tot, dt),
                               "merge_only_seconds": round(tot, 3), "symbols_merged": nsr, "identical_fmd": hashlib.md5(rr.stdout).hexdigest() == hashlib.md5(ra.stdout).hexdigest(),
                               "same_prefix_on_the_gpu": {"merge_path_ms": sa.get("merge_path_ms"), "text_upload_ms": sa.get("text_upload_ms"),
                                                          "speedup_merge_path_incl_upload": round(tot * 1e3 / (sa.get("merge_path_ms", 0) + sa.get("text_upload_ms", 0)), 1) if sa.get("merge_path_ms") else None}}
        if gold:
            out["cpu_baseline"]["recorded_full_run"] = {"reference_seconds": gold.get("reference_seconds"), "reference_merge_only_seconds": gold.get("reference_merge_only_seconds"), "threads": gold.get("reference_threads"),
                                                        "where": "the build container (8 cores), tools/make_golden_mtb.py; not re-timed here"}
    if not keep:
        for f in files:
            os.unlink(f)
        try:
            os.rmdir(tmp)
        except OSError:
            pass
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--genome-len", type=int, default=4400000)
    ap.add_argument("--div", type=float, default=0.001)
    ap.add_argument("--walker-step", type=int, default=384, help="text distance between LF walkers handed to the engine (entry points that take them)")
    ap.add_argument("--entry", choices=["plain", "rows", "text"], default="plain",
                    help="plain: rb3gpu_merge_plain_dev, the reference's signature (default, = value); rows: + sampled inverse suffix array; text: + inverse suffix array (the CLI's path)")
    ap.add_argument("--plain-abi", action="store_true", help="same as --entry plain")
    ap.add_argument("--row-words", action="store_true", help="same as --entry rows")
    ap.add_argument("--mode", choices=["interval", "partition", "replicated"], default=None,
                    help="how the work is split over the GPUs (ropebwt3_amd/multi.py); N>1 default: interval (north_star).  With N=1, --mode interval runs the same sharded step on one GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-aux", action="store_true", help="skip the auxiliary measurements (entry points, reads regime, large index)")
    ap.add_argument("--no-target", action="store_true", help="skip the mtb152 end-to-end leg")
    ap.add_argument("--aux-reads", type=int, default=100000)
    ap.add_argument("--large-index", type=int, default=1 << 30, help="symbols of the index of the large-index leg (0: skip)")
    ap.add_argument("--only", choices=["large", "reads", "target"], default=None, help="run one auxiliary leg alone and print its JSON (profiling)")
    ap.add_argument("--mtb", type=int, default=152, help="genomes of the target-workload leg")
    ap.add_argument("--mtb-ref-prefix", type=int, default=6, help="files the reference binary is timed on (cpu_baseline of the target workload)")
    args = ap.parse_args()
    if args.plain_abi:
        args.entry = "plain"
    if args.row_words:
        args.entry = "rows"

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world

    import torch
    from ropebwt3_amd import Rb3Gpu, host
    from tests import util

    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: torch.cuda.is_available() is False and the engine has no CPU fallback")
    if world > 1 or args.mode is not None:
        from ropebwt3_amd import multi
        args.mode = args.mode or "interval"
        return multi.bench_main(args, rank, local_rank, world)
    torch.cuda.set_device(local_rank)
    if args.only:
        mk = lambda: Rb3Gpu(device=local_rank, verbose=1)
        print(json.dumps(large_index_regime(mk, args.large_index, 1000000) if args.only == "large" else reads_regime(mk, args.aux_reads) if args.only == "reads" else
                         target_workload(args.mtb, 4400000, 0 if args.no_cpu_baseline else args.mtb_ref_prefix)), flush=True)
        return

    t0 = time.time()
    g0, gs = gen_genomes(args.genome_len, args.div, 1, [2])
    b1 = host.build_bwt(util.make_text([g0]))
    text2 = util.make_text(gs)
    b2, w_rows = host.build_bwt_walkers(text2.copy(), args.walker_step)
    w_text = host.walkers_text(text2, args.walker_step)
    log("inputs: B1 %d symbols, B2 %d symbols; host suffix sorting %.1f s (not timed)" % (b1.size, b2.size, time.time() - t0))

    h = Rb3Gpu(device=local_rank, verbose=1)
    h.from_plain(b1)
    d_b2 = h.dev_upload(b2)
    d_b2s, d_tw = h.sort_text(text2)   # the batch as the GPU suffix sorter leaves it in HBM: BWT + inverse suffix array
    assert np.array_equal(h.dev_download(d_b2s, b2.size), b2), "GPU and host suffix sorters disagree"

    steps = {"plain": lambda commit=False: h.merge_plain_dev(d_b2, b2.size, commit=commit),
             "rows": lambda commit=False: h.merge_plain_dev_walkers(d_b2, b2.size, w_rows, commit=commit),
             "text": lambda commit=False: h.merge_text_dev(d_b2s, d_tw, b2.size, w_text, commit=commit)}
    names = {"plain": "rb3gpu_merge_plain_dev (the reference's signature rb3_fmi_merge_plain(r, len, bwt): BWT only; the text-regular walker list is made on the device inside the step)",
             "rows": "rb3gpu_merge_plain_dev_walkers (BWT + inverse suffix array sampled every %d text positions, as a host suffix sorter has it; %d walkers)" % (args.walker_step, len(w_rows)),
             "text": "rb3gpu_merge_text_dev (BWT + inverse suffix array of the batch, both as the GPU suffix sorter leaves them in HBM: the CLI's path; %d walkers)" % len(w_text)}
    step = steps[args.entry]

    def barrier():
        h.sync()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    h.stats_reset()
    barrier()
    t = time.perf_counter()
    for _ in range(args.steps):
        step()
    h.sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    st = h.stats()

    def timed(fn, reps=10):
        fn()
        h.sync()
        h.stats_reset()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        h.sync()
        e = (time.perf_counter() - t) / reps
        s = h.stats()
        return e, s["ms_chain"] / max(1, s["n_rank_launches"]), s["n_fallbacks"]

    aux = {}
    if not args.no_aux:
        for k in ("plain", "rows", "text"):
            e, c, fb = timed(steps[k])
            aux[names[k]] = {"ms_per_step": round(e * 1e3, 4), "Gbp/s": round(b2.size / e / 1e9, 3), "k_chain_ms": round(c, 4), "rank_phase_fallbacks": int(fb)}
        # the same signature with the batch in HOST memory (rb3gpu_merge_plain commits, so it is timed once, on a scratch handle
        # whose pinned staging buffers exist already): PCIe-inclusive, never `value`
        hs = Rb3Gpu(device=local_rank, verbose=1)
        hs.from_plain(b1)
        hs.mg_rank_plain(b2)
        hs.stats_reset()
        t = time.perf_counter()
        hs.merge_plain(b2)
        e = time.perf_counter() - t
        ss = hs.stats()
        hs.close()
        aux["rb3gpu_merge_plain (the same signature with the batch in HOST memory: the PCIe copy of %d bytes, through pinned staging buffers, is inside the call; one call)" % b2.size] = {
            "ms_per_step": round(e * 1e3, 4), "Gbp/s": round(b2.size / e / 1e9, 3), "h2d_ms": round(ss["ms_h2d"], 4), "note": "PCIe-inclusive; never `value`"}

    # one committed merge, to make sure the timed path produces a consistent index
    step(commit=True)
    acc = h.get_acc()
    assert acc[6] == b1.size + b2.size

    sym_per_step = b2.size
    value = sym_per_step * args.steps / dt / 1e9
    ms_chain = st["ms_chain"] / max(1, st["n_rank_launches"])
    out = {
        "metric": "Gbp/s indexed (build merge)", "value": round(value, 6), "unit": "Gbp/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64", "data": "synthetic",
        "config": {"workload": "cfg2-synthetic-mtb1: merge G1 = G0 + 0.1%% substitutions (%d bp, both strands, %d symbols, 2 strings) into the index of G0 (%d symbols)" % (args.genome_len, b2.size, b1.size),
                   "symbols_per_step_per_gpu": int(b2.size), "index_symbols": int(b1.size), "parallelism": "single GPU",
                   "entry_point": names[args.entry], "inputs_resident_in_hbm": True,
                   "lf_steps_per_step": int(st["n_lf_steps"] // max(1, args.steps)), "rank_phase_fallbacks": int(st["n_fallbacks"]),
                   "chain_launches_per_step": round(st["n_rank_launches"] / max(1, args.steps), 2)},
        "phases_ms_per_step": {"lf": round(st["ms_lf"] / args.steps, 4), "rank": round(st["ms_rank"] / args.steps, 4),
                               "rebuild": round(st["ms_build"] / args.steps, 4)},
        "roofline": chain_roofline(b2.size, ms_chain, args.entry, load_pmc_traffic("k_chain"),
                                   "latency-bound regime (2 strings = 2 dependent chains in the reference; here ~23-34 k walkers of a few hundred steps): per step one 128-B slot line and one 8-B record at random rows; "
                                   "`achieved` prices SURVEY 8(d)'s 208 B/step, `achieved_engine` the bytes this engine issues, `achieved_counter` the FETCH_SIZE/WRITE_SIZE traffic of profiles/ over the same duration; "
                                   "the index (4.4 MB) sits in L2/Infinity Cache here -- aux_large_index is the HBM-resident case"),
    }
    if aux:
        out["aux_entry_points"] = aux
    h.dev_free(d_b2)
    h.dev_free(d_b2s)
    h.dev_free(d_tw)
    h.close()
    if not args.no_aux:
        out["aux_reads_regime"] = reads_regime(lambda: Rb3Gpu(device=local_rank, verbose=1), args.aux_reads)
        if args.large_index > 0:
            try:
                out["aux_large_index"] = large_index_regime(lambda: Rb3Gpu(device=local_rank, verbose=1), args.large_index, 1000000)
            except Exception as e:   # (a box with less free memory than the leg needs must not lose the headline)
                out["aux_large_index"] = {"error": repr(e)[:300]}
    if not args.no_target:
        out["target_workload"] = target_workload(args.mtb, 4400000, 0 if args.no_cpu_baseline else args.mtb_ref_prefix)
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(b1, b2)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
