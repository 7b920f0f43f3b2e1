#!/usr/bin/env python3
"""bench.py -- merge-path throughput of the MI355X engine (BASELINE.json metric: "Gbp/sec indexed (build merge)").

HEADLINE (value, ms_per_step, config.workload, roofline, cpu_baseline) = BASELINE configs[2], the north_star's target:
**mtb152**, 152 synthetic M. tuberculosis-like genomes of 4.4 Mbp (tools/gen_mtb.py: star phylogeny, 0.1 % substitutions +
10 indels each), one genome per batch as the reference is run = 151 merge rounds, 1,328,837,368 symbols merged.

A STEP is one complete pass of the hot path over that input: the merge path of the whole build as SURVEY 8(d) defines the
metric -- per round the H2D copy of the batch (its nt6 text, from page-locked memory) + LF array + all LF walkers (rank) +
interleave/rebuild, summed over the 151 rounds; suffix sorting of the batch (the reference's libsais call, excluded by the
metric's definition) runs on the GPU between the two timed parts of a round and is not counted; file I/O happens before the
timed region.  Every timed part is bracketed by host clocks with the device idle on both sides (the upload returns after its
stream synchronisation, a merge ends with its one synchronisation).  After the last step the index is packed into an .fmd
on the GPU and its md5 is compared with the one the UNMODIFIED REFERENCE produced for the same 152 files
(tests/golden/MANIFEST.json, tools/make_golden_mtb.py): `fmd_identical_to_reference`.

The same JSON line also carries: `roofline` (dominant kernel k_chain<mixed>: 208 B x LF steps / its HIP-event time; traffic
from the committed rocprofv3 --pmc passes), `roofline_path` (SURVEY 8(d)'s whole-path formula), `cpu_baseline` (the unmodified
reference binary timed here on a stated prefix of the same files), and auxiliary legs, each labelled: the whole build through
the CLI (sorting overlapped, wall clock), BASELINE configs[1] (one genome into one: the former headline), the reads regime and
the HBM-resident large index.

N > 1 (python bench.py --gpus N starts its N ranks itself; under torch.distributed.run it uses the ranks it is given): the SAME
build, partitioned -- rank r builds the index of its contiguous slice of the 152 genomes, then the slices are combined by a
binary tree of whole-index merges over RCCL (ropebwt3_amd/multi.py); one step = the whole partitioned build, same md5 gate,
`scaling` = "strong" (the job is fixed: one mtb152 index).

    python bench.py [--gpus N] [--steps K] [--warmup W]
"""
import argparse
import hashlib
import json
import os
import re
import shutil
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


def load_pmc_traffic(kernel):
    """HBM bytes per launch of a kernel from the committed rocprofv3 --pmc passes (profiles/), if any."""
    for fn in ("r6_pmc_%s.json" % kernel, "r5_pmc_%s.json" % kernel, "r4_pmc_%s.json" % kernel, "r3_pmc_%s.json" % kernel, "r2_pmc_%s.json" % kernel, "r1_pmc_%s.json" % kernel):
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
    d, d_tw, d_sa = h.sort_text_sa(t2)          # (+ the suffix array: records in text order, gathered by the validation pass)
    h.merge_text_dev(d, d_tw, t2.size, 2 * n_reads, commit=False, d_sa=d_sa)
    h.stats_reset()
    reps = 3
    t = time.perf_counter()
    for _ in range(reps):
        h.merge_text_dev(d, d_tw, t2.size, 2 * n_reads, commit=False, d_sa=d_sa)
    dt = (time.perf_counter() - t) / reps
    st_ = h.stats()
    ms_chain = st_["ms_chain"] / reps
    out = {"workload": "large index: merge %d x 150 bp reads (both strands, %d symbols) into the index of a random genome, %d symbols = %.0f MB of slots in HBM (index built in %.1f s incl. GPU suffix sorting)" % (n_reads, t2.size, n_index, st_["bytes_index"] / 1e6, t_idx),
           "value": round(t2.size / dt / 1e9, 4), "unit": "Gbp/s", "ms_per_step": round(dt * 1e3, 3),
           "phases_ms_per_step": {"lf": round(st_["ms_lf"] / reps, 3), "rank": round(st_["ms_rank"] / reps, 3), "rebuild": round(st_["ms_build"] / reps, 3)},
           "rebuild_streaming": {"bytes_per_step": int(st_["bytes_rebuild"] // reps), "GB/s": round(st_["bytes_rebuild"] / max(1e-9, st_["ms_build"]) / 1e6, 1),
                                 "frac": round(st_["bytes_rebuild"] / max(1e-9, st_["ms_build"]) / 1e6 / HBM_PEAK_GBS, 4)},
           "roofline": dict(chain_roofline(t2.size, st_["ms_rank"] / reps, "text", None,
                                           "every slot read comes from HBM (index >> 256 MB of L2 + Infinity Cache).  Priced over the RANK PHASE, not k_chain alone: the walkers leave their "
                                           "records in text order (one 64-byte store per 8 steps) and the validation pass gathers them into row order through the batch's suffix array "
                                           "(rb3gpu_merge_text_sa_dev), so the 16 B of row traffic per step of SURVEY 8(d) are spent in k_pos_finalize_check_rows; k_chain alone: k_chain_ms_per_launch. "
                                           "With a record per row (RB3GPU_TREC=0: rounds 1-2 and most of 3) the walk took 16.6-18.2 ms and the phase 18.4-20.7"),
                            kernel="k_chain + k_pos_finalize_check_rows (the rank phase)", k_chain_ms_per_launch=round(ms_chain, 4),
                            traffic_k_chain=load_pmc_traffic("k_chain_large"))}
    h.dev_free(d)
    h.dev_free(d_tw)
    h.dev_free(d_sa)
    h.close()
    return out


def index_8g(n_hap, device=0):
    """An index of more than 2^32 symbols (the regime of BASELINE configs[3]/[4]): 24 haplotypes of 180 Mbp, one per
    merge round, to 8.64 G symbols; then 1 M reads into it.  tools/big_index.py (in process, through the C ABI)."""
    from tools import big_index
    t0 = time.time()
    h, srt, rounds, base = big_index.build(n_hap, 180000000, log=log, device=device)
    try:
        rd = big_index.reads_into(h, base, 1000000)
        return big_index.summary(rounds, rd, n_hap, 180000000, time.time() - t0)
    finally:
        srt.close()
        h.close()


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



# ---------------------------------------------------------------------------------------------
# the headline workload: mtb152, in process through the C ABI
# ---------------------------------------------------------------------------------------------

MTB_L = 4400000
def protect_stdout():
    """The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a version banner when a communicator is made, and a C
    library's buffered output comes out when the process ends, i.e. BEHIND the line): from here on file descriptor 1 is stderr for everybody,
    and emit_json() writes the line to the real stdout (its descriptor travels in the environment: ropebwt3_amd/multi.py imports this file as
    a second module)."""
    if "RB3_BENCH_STDOUT_FD" not in os.environ:
        sys.stdout.flush()
        os.environ["RB3_BENCH_STDOUT_FD"] = str(os.dup(1))
        os.dup2(2, 1)


def emit_json(obj):
    line = (json.dumps(obj) + "\n").encode()
    fd = os.environ.get("RB3_BENCH_STDOUT_FD")
    if fd is None:
        sys.stdout.write(line.decode()), sys.stdout.flush()
    else:
        os.write(int(fd), line)


WALKER_STEP = 0      # 0: rb3gpu_walker_step (as many walkers as k_chain keeps resident: 230 text positions for a 4.4 Mbp genome on an MI355X)


def mtb_manifest(K, L):
    man = json.load(open(os.path.join(ROOT, "tests", "golden", "MANIFEST.json"))).get("mtb_star", {})
    return man.get("prefixes", {}).get(str(K)) if L == man.get("genome_len") else None


def mtb_files(K, L, tmp=None):
    from tools import gen_mtb
    tmp = tmp or tempfile.mkdtemp(prefix="rb3_mtb_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    t = time.time()
    files = gen_mtb.generate(K, L, tmp)
    return tmp, files, time.time() - t


def load_batches(files, pinned=True, step=WALKER_STEP, device=0, host_walkers=False):
    """every file through the CLI's own reader (rb3h_seq_read: nt6, both strands, sentinels; io.c:104-125), one batch per
    file, into page-locked memory (what `ropebwt3-amd build` does with rb3gpu_pinned_alloc) + the walker list of each batch
    (one walker per string + one per rb3gpu_walker_step text positions -- 230 for these genomes --, rb3h_walkers_text).  Not timed: file I/O is outside the metric."""
    from ropebwt3_amd import PinnedArray, host, walker_step
    load_batches.host_walkers = host_walkers
    texts, walkers, keep = [], [], []
    load_batches.walker_seconds = 0.0   # host time spent making walker lists (rb3h_walkers_text; --host-walkers only), all batches
    load_batches.pairs_seconds = 0.0    # ... and finding where the records start (rb3h_strand_pairs: what the forward-strand upload needs; the CLI's reader knows them as it parses)
    for fn in files:
        parts = list(host.read_batches(fn, False, 1 << 40))
        n_seq = sum(k for k, _ in parts)
        t = parts[0][1] if len(parts) == 1 else np.concatenate([x for _, x in parts])
        if pinned:
            pa = PinnedArray(t.size)
            pa.array[:] = t
            keep.append(pa)
            t = pa.array
        texts.append(t)
        wstep = step if step > 0 else walker_step(device, t.size, n_seq)
        tw0 = time.perf_counter()
        sp = host.strand_pairs(t, n_seq)
        load_batches.pairs_seconds += time.perf_counter() - tw0
        if not load_batches.host_walkers:   # the walker list is made on the device, inside the timed merge call (rb3gpu_merge_text_step_dev): the host only says how many strings and what spacing
            walkers.append(((int(n_seq), int(wstep)), sp))
            continue
        tw0 = time.perf_counter()
        w = host.walkers_text(t, wstep)
        load_batches.walker_seconds += time.perf_counter() - tw0
        if pinned:  # the list goes to the device inside the merge call: from page-locked memory that is one DMA, no staging copy on the host
            pw = PinnedArray(w.nbytes)
            wv = pw.array.view(np.int64).reshape(w.shape)
            wv[:] = w
            keep.append(pw)
            w = wv
        walkers.append((w, sp))   # (+ where the records start: the CLI's sorter thread finds them the same way)
    return texts, walkers, keep


class BuildLoop:
    """`ropebwt3-amd build` of a list of batches in this process, one call of the C ABI per stage so that the stages can be
    timed apart: rb3gpu_sorter_upload (H2D, timed) -> rb3gpu_sorter_sort_uploaded (suffix sorting: not part of the metric) ->
    rb3gpu_from_plain_dev for the first batch / rb3gpu_merge_text_dev for every later one (timed)."""

    def __init__(self, device):
        from ropebwt3_amd import Rb3Gpu, Sorter
        self.h = Rb3Gpu(device=device, verbose=int(os.environ.get("RB3_BENCH_VERBOSE", "1")))   # (2: the engine's warnings, e.g. why a merge was redone)
        self.srt = Sorter(device)
        self.fwd_upload = True
        self.overlap = True      # the H2D copy of batch i + 1 runs beside the merge of batch i (--serial-h2d: one after the other)

    def run(self, texts, walkers, first_is_index=True, reference_signature=False):
        """returns (seconds H2D, seconds merge, seconds sort, symbols merged, wall seconds) of one build.
        Serial: upload(i) -> sort(i) -> merge(i), the upload timed from call to completion.  Overlapped (the default; what the CLI's
        sorter thread does, and build.c:203-239 with its reader): the copies of batch i + 1 are QUEUED before merge(i) is called
        and run on the copy engine beside its kernels; H2D seconds = the time to queue them + whatever the copy engine still needs
        when the merge call has returned (rb3gpu_sorter_upload_end, called right behind it) -- nothing of the copy can hide in the
        suffix sorting, which is not part of the metric."""
        h, srt = self.h, self.srt
        t_h2d = t_mrg = t_sort = 0.0
        nsym = 0
        w0 = time.perf_counter()
        n = len(texts)
        overlap = self.overlap and self.fwd_upload and all(p is not None for _, p in walkers)
        uploaded = -1            # the batch whose text is on the device

        def upload(i, begin_only=False):
            t, (_, pairs) = texts[i], walkers[i]
            if begin_only:
                srt.upload_fwd_begin(t, pairs)
            elif pairs is not None and self.fwd_upload:
                srt.upload_fwd(t, pairs)                    # forward strands over PCIe, reverse complements made on the device
            else:
                srt.upload(t)                               # returns after the stream synchronisation

        for i, (t, (w, pairs)) in enumerate(zip(texts, walkers)):
            a = time.perf_counter()
            if uploaded != i:
                upload(i)
            b = time.perf_counter()
            d_bwt, d_tw, d_sa = srt.sort_uploaded_sa(t.size)   # (synchronous; the suffix array rides along for the engine to use where it pays)
            c = time.perf_counter()
            if i == 0 and first_is_index:
                if overlap and n > 1:
                    upload(1, True), srt.upload_end()       # (timed below like every other batch's: queue + wait, here with nothing beside it)
                    uploaded = 1
                    t_h2d += time.perf_counter() - c
                h.from_plain_dev(d_bwt, t.size)             # rb3_enc_plain2fmr: the first batch is encoded, not merged
            else:
                if overlap and i + 1 < n:
                    upload(i + 1, True)                     # queued on the sorter's stream: returns at once
                    uploaded = i + 1
                c1 = time.perf_counter()
                if reference_signature:   # the arguments of rb3_fmi_merge_plain (fm-index.c:279): the partial BWT and nothing else
                    h.merge_plain_dev(d_bwt, t.size, commit=True)
                elif isinstance(w, tuple):   # (strings, spacing): the walker list is made on the device, inside this call
                    h.merge_text_step_dev(d_bwt, d_tw, t.size, w[0], w[1], commit=True, d_sa=d_sa)   # one synchronisation, at its end
                else:
                    h.merge_text_dev(d_bwt, d_tw, t.size, w, commit=True, d_sa=d_sa)
                d = time.perf_counter()
                if overlap and i + 1 < n:
                    srt.upload_end()                        # what the copy engine still has to do shows up here
                e = time.perf_counter()
                t_h2d += (b - a) + (c1 - c) + (e - d)
                t_mrg += d - c1
                nsym += t.size
            t_sort += c - b
            srt.release(d_bwt)
        return t_h2d, t_mrg, t_sort, nsym, time.perf_counter() - w0

    def fmd_md5(self):
        from ropebwt3_amd import host
        data = host.fmd_bytes_from_words(self.h.export_fmd_words(), self.h.get_acc())
        return hashlib.md5(data).hexdigest(), len(data)

    def close(self):
        self.srt.close()
        self.h.close()


def reference_prefix(files, n_ref, K):
    """cpu_baseline of the headline: the UNMODIFIED reference binary (oracle/_ref/ropebwt3, built by oracle/Makefile from the
    sources under /root/reference; it travels with the repository) on the first n_ref of the same files, merge-only seconds by
    its own timers, next to this engine on the same prefix."""
    from ropebwt3_amd import _build
    ref = os.path.join(ROOT, "oracle", "_ref", "ropebwt3")
    if n_ref < 2 or not os.path.exists(ref):
        return None
    cores = os.cpu_count() or 1
    nthr = min(cores, 64)
    t = time.time()
    rr = subprocess.run([ref, "build", "-d", "-t%d" % nthr] + files[:n_ref], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
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
    ra = subprocess.run([_build.BIN_CLI, "build", "-d"] + files[:n_ref], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    sa = parse_cli_stats(ra.stderr.decode())
    gold = mtb_manifest(K, MTB_L)
    out = {"value": round(nsr / tot / 1e9, 6) if tot > 0 else None, "unit": "Gbp/s", "cores": cores, "kind": "reference",
           "sample": "`ropebwt3 build -d -t%d` (oracle/_ref, unmodified) on the first %d of the %d files: merge-only seconds by its own timers ('inserted' minus 'constructed partial BWT', summed over %d rounds) %.2f s of %.1f s; rb3_fmi_merge_plain has one chain per string, i.e. 2 threads of work per round whatever -t says" % (nthr, n_ref, K, n_ref - 1, tot, dt),
           "merge_only_seconds": round(tot, 3), "symbols_merged": nsr, "identical_fmd": hashlib.md5(rr.stdout).hexdigest() == hashlib.md5(ra.stdout).hexdigest(),
           "same_prefix_on_the_gpu": {"merge_path_ms_incl_h2d": round(sa.get("merge_path_ms", 0) + sa.get("text_upload_ms", 0), 3) if sa.get("merge_path_ms") else None,
                                      "speedup_merge_path": round(tot * 1e3 / (sa.get("merge_path_ms", 0) + sa.get("text_upload_ms", 0)), 1) if sa.get("merge_path_ms") else None}}
    if gold:
        out["recorded_full_run"] = {"reference_seconds": gold.get("reference_seconds"), "reference_merge_only_seconds": gold.get("reference_merge_only_seconds"), "threads": gold.get("reference_threads"),
                                    "where": "the build container (8 cores), tools/make_golden_mtb.py; not re-timed here"}
    return out


def reference_many_chains(files, n_first, n_batch):
    """SURVEY 8(d) config 3(b) -- the reference at its BEST on this workload: the same genomes concatenated so that ONE merge round holds 2 * n_batch
    chains (rb3_fmi_merge_plain runs one chain per string over kt_for, fm-index.c:217-224): file 1 = the first n_first genomes (the first batch: no
    merge), file 2 = the next n_batch genomes, `-m` large enough for each file to be one batch.  Merge-only seconds by the reference's own timers."""
    ref = os.path.join(ROOT, "oracle", "_ref", "ropebwt3")
    if not os.path.exists(ref) or len(files) < n_first + n_batch:
        return None
    from ropebwt3_amd import _build
    cores = os.cpu_count() or 1
    nthr = min(cores, 2 * n_batch)
    tmp = tempfile.mkdtemp(prefix="rb3_3b_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        fa, fb = os.path.join(tmp, "first.fa"), os.path.join(tmp, "batch.fa")
        for fn, part in ((fa, files[:n_first]), (fb, files[n_first:n_first + n_batch])):
            with open(fn, "wb") as fp:
                for f in part:
                    fp.write(open(f, "rb").read())
        t = time.time()
        rr = subprocess.run([ref, "build", "-d", "-m8g", "-t%d" % nthr, fa, fb], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
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
        ra = subprocess.run([_build.BIN_CLI, "build", "-d", "-m8g", fa, fb], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        sa = parse_cli_stats(ra.stderr.decode())
        gms = (sa.get("merge_path_ms", 0) + sa.get("text_upload_ms", 0)) if sa.get("merge_path_ms") else None
        return {"value": round(nsr / tot / 1e9, 6) if tot > 0 else None, "unit": "Gbp/s", "cores": cores, "kind": "reference",
                "sample": "`ropebwt3 build -d -m8g -t%d first.fa batch.fa` (oracle/_ref, unmodified): %d genomes as the first batch, the next %d concatenated as ONE batch = one merge round of %d chains on %d threads; merge-only seconds by its own timers %.2f s of %.1f s wall" % (nthr, n_first, n_batch, 2 * n_batch, nthr, tot, dt),
                "merge_only_seconds": round(tot, 3), "symbols_merged": nsr, "chains_per_round": 2 * n_batch,
                "identical_fmd": rr.returncode == 0 and hashlib.md5(rr.stdout).hexdigest() == hashlib.md5(ra.stdout).hexdigest(),
                "same_files_on_the_gpu": {"merge_path_ms_incl_h2d": round(gms, 3) if gms else None, "speedup_merge_path": round(tot * 1e3 / gms, 1) if gms else None}}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def cli_build(files, K, gold):
    """the whole build through the CLI (reader thread, GPU sorter thread and merges overlapped; wall clock incl. process start,
    file reading and FMD writing) + the same files re-batched"""
    from ropebwt3_amd import _build
    best = None
    for _ in range(2):   # the first run pays for the page cache and the HIP start-up; report the second
        t = time.time()
        r = subprocess.run([_build.BIN_CLI, "build", "-d"] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        wall = time.time() - t
        if r.returncode != 0:
            return {"error": r.stderr.decode()[-400:]}
        best = (wall, r)
    wall, r = best
    md5 = hashlib.md5(r.stdout).hexdigest()
    st = parse_cli_stats(r.stderr.decode())
    out = {"command": "ropebwt3-amd build -d g000.fa ... g%03d.fa" % (K - 1), "build_wall_s": round(wall, 3), "fmd_md5": md5, "fmd_identical_to_reference": (md5 == gold["fmd_md5"]) if gold else None,
           "merge_path_ms": st.get("merge_path_ms"), "text_upload_ms": st.get("text_upload_ms"), "phases_ms": {"lf": st.get("lf_ms"), "rank": st.get("rank_ms"), "rebuild": st.get("rebuild_ms")},
           "suffix_sorting_ms_overlapped": st.get("sort_ms"), "k_chain_ms": st.get("chain_ms"), "index_mb": st.get("index_mb"),
           "batches_sorted_on_gpu": st.get("batches_gpu"), "batches_sorted_on_host": st.get("batches_host"),
           "note": "merges run beside the sorter thread's kernels here, so the per-phase times are longer than in the headline, whose stages run one at a time"}
    t = time.time()
    rb = subprocess.run([_build.BIN_CLI, "build", "-d", "--rebatch", "-m80m"] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    wall_b = time.time() - t
    if rb.returncode == 0:
        sb = parse_cli_stats(rb.stderr.decode())
        out["rebatched"] = {"command": "ropebwt3-amd build -d --rebatch -m80m (9-10 genomes per batch)", "build_wall_s": round(wall_b, 3), "merge_rounds": sb.get("chain_launches"),
                            "merge_path_ms": sb.get("merge_path_ms"), "text_upload_ms": sb.get("text_upload_ms"), "fmd_identical": hashlib.md5(rb.stdout).hexdigest() == md5}
    return out


def cfg2_step(local_rank, steps, warmup, genome_len, div):
    """BASELINE configs[1] (the headline of rounds 1-2): one genome merged into the index of one, through the entry point with
    the reference's signature (BWT only) and through the one the CLI uses (BWT + inverse suffix array from the GPU sorter)"""
    from ropebwt3_amd import Rb3Gpu, host, walker_step
    from tests import util
    g0, gs = gen_genomes(genome_len, div, 1, [2])
    b1 = host.build_bwt(util.make_text([g0]))
    text2 = util.make_text(gs)
    w_text = host.walkers_text(text2, WALKER_STEP if WALKER_STEP > 0 else walker_step(0, text2.size, 2))
    h = Rb3Gpu(device=local_rank, verbose=1)
    h.from_plain(b1)
    d_b2s, d_tw = h.sort_text(text2)
    out = {"workload": "cfg2-synthetic-mtb1: merge G1 = G0 + 0.1%% substitutions (%d bp, both strands, %d symbols, 2 strings) into the index of G0 (%d symbols); batch resident in HBM, commit=0" % (genome_len, text2.size, b1.size)}
    for name, fn in (("rb3gpu_merge_plain_dev (reference's signature: BWT only, text-order words made on the device)", lambda: h.merge_plain_dev(d_b2s, text2.size, commit=False)),
                     ("rb3gpu_merge_text_dev (BWT + inverse suffix array as the GPU sorter leaves them: the CLI's path)", lambda: h.merge_text_dev(d_b2s, d_tw, text2.size, w_text, commit=False))):
        for _ in range(warmup):
            fn()
        h.sync()
        h.stats_reset()
        t = time.perf_counter()
        for _ in range(steps):
            fn()
        h.sync()
        e = (time.perf_counter() - t) / steps
        s = h.stats()
        out[name] = {"ms_per_step": round(e * 1e3, 4), "Gbp/s": round(text2.size / e / 1e9, 3), "k_chain_ms": round(s["ms_chain"] / max(1, s["n_rank_launches"]), 4),
                     "frac_of_208B_roofline": round(ALGO_BYTES_PER_STEP * text2.size / (s["ms_chain"] / max(1, s["n_rank_launches"]) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "rank_phase_fallbacks": int(s["n_fallbacks"])}
    hs = Rb3Gpu(device=local_rank, verbose=1)   # host buffer + PCIe copy inside the call (pageable memory, staged): one call on a warm handle
    hs.from_plain(b1)
    b2 = h.dev_download(d_b2s, text2.size)
    hs.mg_rank_plain(b2)
    hs.stats_reset()
    t = time.perf_counter()
    hs.merge_plain(b2)
    e = time.perf_counter() - t
    out["rb3gpu_merge_plain (host buffer: the PCIe copy of %d bytes from pageable memory is inside the call)" % b2.size] = {"ms": round(e * 1e3, 4), "h2d_ms": round(hs.stats()["ms_h2d"], 4)}
    hs.close()
    h.dev_free(d_b2s)
    h.dev_free(d_tw)
    h.close()
    return out


def self_launch(args):
    """python bench.py --gpus N without a launcher: start the N ranks here (torch.distributed.run, one process per GPU)"""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["RB3_BENCH_SELF_LAUNCHED"] = "1"
    env.pop("RB3_BENCH_STDOUT_FD", None)   # a descriptor number of THIS process means nothing in the ranks (close_fds): each rank saves its own stdout
    sys.stdout.flush()
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3, help="timed steps; a step = the merge path of one whole mtb152 build (151 merge rounds)")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mtb", type=int, default=152, help="genomes of the headline workload (152 = BASELINE configs[2]; 24 has a golden md5 too)")
    ap.add_argument("--genome-len", type=int, default=MTB_L)
    ap.add_argument("--div", type=float, default=0.001)
    ap.add_argument("--mode", choices=["partition", "interval", "replicated"], default=None,
                    help="N > 1: how the work is split over the GPUs (ropebwt3_amd/multi.py); default partition = the same mtb152 build, partitioned + tree merge")
    ap.add_argument("--sh-driver", choices=["rccl", "torch", "python", "gloo"], default=None, help="--mode interval: what runs the lock-step loop and carries the states: rb3gpu_sh_merge over the library's RCCL communicator (default), over torch.distributed callbacks, or the loop in Python (rounds 1-3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--peer-rounds", type=int, default=100000, help="reads of the aux_interval_peer_rounds leg (0: skip it)")
    ap.add_argument("--no-aux", action="store_true", help="skip the auxiliary legs (CLI build, configs[1], reads regime, large index)")
    ap.add_argument("--no-pinned", action="store_true", help="batches in pageable host memory (staged upload)")
    ap.add_argument("--serial-h2d", action="store_true", help="upload every batch before its own sort with nothing beside it (round 3's first definition) instead of beside the merge of the batch before")
    ap.add_argument("--full-upload", action="store_true", help="copy both strands of every batch over PCIe (rb3gpu_sorter_upload) instead of the forward strands only")
    ap.add_argument("--aux-reads", type=int, default=100000)
    ap.add_argument("--large-index", type=int, default=1 << 30, help="symbols of the index of the large-index leg (0: skip)")
    ap.add_argument("--index-8g", type=int, default=24, help="haplotypes (360 M symbols each) of the leg with an index beyond 2^32 symbols (0: skip)")
    ap.add_argument("--scale", choices=["reads", "hap"], default=None, help="run one of the scale scripts now (tools/r6/scale_reads.sh: 60 M reads, 18 G symbols; tools/r6/scale_hap.sh: 4 haplotypes x 3.1 Gbp) and print its JSON")
    ap.add_argument("--only", choices=["large", "reads", "cfg2", "cli", "headline", "8g"], default=None, help="run one leg alone and print its JSON (profiling)")
    ap.add_argument("--mtb-ref-prefix", type=int, default=6, help="files the reference binary is timed on (cpu_baseline)")
    ap.add_argument("--mtb-3b-first", type=int, default=4, help="cpu_baseline_config3b: genomes in the first batch of the reference's many-chain run")
    ap.add_argument("--mtb-3b-batch", type=int, default=16, help="cpu_baseline_config3b: genomes concatenated into the ONE merged batch (0 = skip)")
    ap.add_argument("--walker-step", type=int, default=WALKER_STEP)
    ap.add_argument("--host-walkers", action="store_true", help="make the walker lists on the host before the timed steps (rb3h_walkers_text, rounds 2-4) instead of on the device inside the merge call")
    args = ap.parse_args()
    if args.scale:   # (a recorded-run script, run now: its JSON object is the line)
        r = subprocess.run(["bash", os.path.join(ROOT, "tools", "r6", "scale_reads.sh" if args.scale == "reads" else "scale_hap.sh")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT)
        lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
        print(lines[-1] if lines else json.dumps({"error": r.stderr.decode()[-400:]}))
        return

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # (decided BEFORE file descriptor 1 is redirected: the ranks must inherit the real stdout, and each of them protects its own)
    if args.gpus > 1 and world == 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args))
    protect_stdout()
    if world > 1:
        args.gpus = world
    if os.environ.get("RB3_BENCH_LAUNCH_SELFTEST"):   # tests/test_cpu_host.py: the launch + stdout plumbing alone, no GPU touched
        if rank == 0:
            print("noise a library would write to stdout")   # (goes to stderr: descriptor 1 is protected)
            emit_json({"selftest": "launch", "n_gpus": args.gpus, "world": world})
        return

    import torch
    from ropebwt3_amd import Rb3Gpu

    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: torch.cuda.is_available() is False and the engine has no CPU fallback")
    if world > 1 or args.mode is not None:
        from ropebwt3_amd import multi
        args.mode = args.mode or "partition"
        return multi.bench_main(args, rank, local_rank, world)
    torch.cuda.set_device(local_rank)
    mk = lambda: Rb3Gpu(device=local_rank, verbose=1)
    if args.only == "8g":
        emit_json(index_8g(args.index_8g, local_rank))
        return
    if args.only in ("large", "reads", "cfg2"):
        emit_json(large_index_regime(mk, args.large_index, 1000000) if args.only == "large" else reads_regime(mk, args.aux_reads) if args.only == "reads" else
                         cfg2_step(local_rank, 10, 3, args.genome_len, args.div))
        return

    K, L = args.mtb, args.genome_len
    gold = mtb_manifest(K, L)
    tmp, files, t_gen = mtb_files(K, L)
    if args.only == "cli":
        emit_json(cli_build(files, K, gold))
        return
    t0 = time.time()
    texts, walkers, keep = load_batches(files, pinned=not args.no_pinned, step=args.walker_step, device=local_rank, host_walkers=args.host_walkers)
    nsym_all = int(sum(t.size for t in texts))
    log("mtb%d: %d files generated in %.1f s, read into %s memory in %.1f s (%d symbols)" % (K, K, t_gen, "pageable" if args.no_pinned else "page-locked", time.time() - t0, nsym_all))

    bl = BuildLoop(local_rank)
    bl.fwd_upload = not args.full_upload
    bl.overlap = not args.serial_h2d and not args.full_upload and not args.no_pinned
    for _ in range(args.warmup):
        bl.run(texts, walkers)
    bl.h.sync()
    torch.cuda.synchronize()
    bl.h.stats_reset()
    tot_h2d = tot_mrg = tot_sort = tot_wall = 0.0
    nsym = 0
    for _ in range(args.steps):
        a, b, c, n, w = bl.run(texts, walkers)
        tot_h2d, tot_mrg, tot_sort, tot_wall, nsym = tot_h2d + a, tot_mrg + b, tot_sort + c, tot_wall + w, nsym + n
    bl.h.sync()
    torch.cuda.synchronize()
    st = bl.h.stats()
    h2d_serial = None
    if bl.overlap:   # for the record: the same copies with nothing beside them (one more build, not part of the timed steps)
        bl.overlap = False
        h2d_serial = bl.run(texts, walkers)[0]
        bl.overlap = True
    md5, fmd_len = bl.fmd_md5()
    # the same 151 rounds through the REFERENCE'S signature -- rb3gpu_merge_plain_dev gets the partial BWT and nothing else
    # (rb3_fmi_merge_plain, fm-index.c:279), so the walkers of the long strings come from a sparse LF walk of the batch itself and the walk reads
    # row words instead of streaming the inverse suffix array: what not having the sorter's products costs on the headline workload
    refsig = None
    if not args.no_aux:
        try:
            bl.run(texts, walkers, reference_signature=True)
            bl.h.sync()
            bl.h.stats_reset()
            a, b, c, n, w = bl.run(texts, walkers, reference_signature=True)
            bl.h.sync()
            s2 = bl.h.stats()
            md5r, _ = bl.fmd_md5()
            refsig = {"workload": "the headline's 151 merge rounds through rb3gpu_merge_plain_dev (the arguments of rb3_fmi_merge_plain: the partial BWT only; round 6: the text-order words are made on the device from the batch's own sparse LF walk -- phase `lf` -- and the batch then goes through the same text path as the headline)",
                      "ms_per_step": round((a + b) * 1e3, 3), "value": round(n / (a + b) / 1e9, 4), "unit": "Gbp/s", "phases_ms_per_step": {"h2d": round(a * 1e3, 3), "lf": round(s2["ms_lf"], 3), "rank": round(s2["ms_rank"], 3), "k_chain": round(s2["ms_chain"], 3), "rebuild": round(s2["ms_build"], 3)},
                      "rank_phase_fallbacks": int(s2["n_fallbacks"]), "fmd_identical_to_reference": (md5r == gold["fmd_md5"]) if gold else None,
                      "ratio_to_the_headline": round((a + b) / ((tot_h2d + tot_mrg) / args.steps), 3)}
        except Exception as e:
            refsig = {"error": repr(e)[:300]}
    # rb3_fmi_merge (fm-index.c:251-277; `ropebwt3 merge`, and the tree step of `build --gpus N`): the index of the second half of the genomes
    # merged into the index of the first half, both resident in HBM -- rb3gpu_merge_index(a, b): b's BWT expanded to 1 byte per symbol, walked
    # through the reference's signature (VERDICT r4 item 8 asks for a walk that ranks on b's block array instead: not built)
    mrg_idx = None
    if not args.no_aux and K >= 4:
        try:
            half = K // 2
            la, lb = BuildLoop(local_rank), BuildLoop(local_rank)
            la.run(texts[:half], walkers[:half]), lb.run(texts[half:], walkers[half:])
            la.h.sync(), lb.h.sync()
            na, nb = la.h.get_tot(), lb.h.get_tot()
            bytes_ab = la.h.stats()["bytes_index"] + lb.h.stats()["bytes_index"]
            la.h.stats_reset()
            t = time.perf_counter()
            la.h.merge_index(lb.h)
            e = time.perf_counter() - t
            sa_ = la.h.stats()
            md5m, _ = la.fmd_md5()
            mrg_idx = {"workload": "rb3gpu_merge_index: the index of genomes %d..%d (%d symbols) merged into the index of genomes 0..%d (%d symbols), both in HBM" % (half, K - 1, nb, half - 1, na),
                       "ms": round(e * 1e3, 3), "value": round(nb / e / 1e9, 4), "unit": "Gbp/s",
                       "phases_ms": {"lf": round(sa_["ms_lf"], 3), "rank": round(sa_["ms_rank"], 3), "k_chain": round(sa_["ms_chain"], 3), "rebuild": round(sa_["ms_build"], 3)},
                       "operands_bytes": int(bytes_ab), "handle_peak_bytes": int(sa_["bytes_peak"]), "peak_over_operands": round(sa_["bytes_peak"] / max(1, bytes_ab), 1),
                       "rank_phase_fallbacks": int(sa_["n_fallbacks"]), "fmd_identical_to_reference": (md5m == gold["fmd_md5"]) if gold else None}
            la.close(), lb.close()
        except Exception as e:
            mrg_idx = {"error": repr(e)[:300]}
    ident = (md5 == gold["fmd_md5"]) if gold else None
    if ident is False:
        log("ERROR: the .fmd differs from the reference's (md5 %s vs %s)" % (md5, gold["fmd_md5"]))
    S = args.steps
    dt = tot_h2d + tot_mrg
    sym_step = nsym // S
    ms_chain = st["ms_chain"] / max(1, st["n_rank_launches"])
    rows_launch = nsym / max(1, st["n_rank_launches"])
    path_bytes = 217 * nsym + (st["bytes_rebuild"] - 9 * nsym)      # 217 B x len + bytes(B1 old) + bytes(B1 new) per round
    out = {
        "metric": "Gbp/s indexed (build merge)", "value": round(nsym / dt / 1e9, 6), "unit": "Gbp/s",
        "n_gpus": 1, "steps": S, "warmup": args.warmup, "ms_per_step": round(dt / S * 1e3, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "int64", "data": "synthetic",
        "config": {"workload": "cfg3-synthetic-mtb%d: %d genomes of %d bp (star phylogeny, 0.1 %% substitutions + 10 indels each; tools/gen_mtb.py), one genome per batch = %d merge rounds per step, %d symbols merged per step; merge path incl. H2D (SURVEY 8(d))%s" % (K, K, L, K - 1, sym_step, ", the H2D copy of batch i+1 queued beside the merge of batch i" if bl.overlap else ""),
                   "symbols_per_step": int(sym_step), "merge_rounds_per_step": K - 1, "index_symbols_final": nsym_all, "index_mb_final": round(st["bytes_index"] / 1e6, 1), "parallelism": "single GPU",
                   "entry_points": "rb3gpu_sorter_upload_fwd (H2D of the batch: forward strands out of page-locked memory, reverse complements written on the device; --full-upload: rb3gpu_sorter_upload) + rb3gpu_merge_text_step_dev (walker list made on the device + LF + walkers + settle + validation + rebuild, commit=1; --host-walkers: rb3gpu_merge_text_dev with a list from the host); rb3gpu_sorter_sort_uploaded between them is not counted (suffix sorting: excluded by the metric)",
                   "fmd_md5": md5, "fmd_bytes": fmd_len, "fmd_identical_to_reference": ident,
                   "reference_fmd_md5_source": "tests/golden/MANIFEST.json mtb_star/%d (oracle/_ref/ropebwt3 = the unmodified reference, tools/make_golden_mtb.py)" % K if gold else "no golden for this size",
                   "lf_steps_per_step": int(st["n_lf_steps"] // S), "rank_phase_fallbacks": int(st["n_fallbacks"]), "long_settles": int(st["n_long_settles"]),
                   "handle_peak_device_bytes": int(st["bytes_peak"]), "handle_peak_bytes_per_batch_symbol": round((st["bytes_peak"] - st["bytes_index"]) / max(1, sym_step // max(1, K - 1)), 1)},
        "phases_ms_per_step": {"h2d": round(tot_h2d / S * 1e3, 3), "merge_calls": round(tot_mrg / S * 1e3, 3), "lf": round(st["ms_lf"] / S, 3), "rank": round(st["ms_rank"] / S, 3), "k_chain": round(st["ms_chain"] / S, 3),
                               "rebuild": round(st["ms_build"] / S, 3), "host_and_sync_inside_merge_calls": round((tot_mrg * 1e3 - st["ms_lf"] - st["ms_rank"] - st["ms_build"]) / S, 3)},
        "not_counted_ms_per_step": {"suffix_sorting_on_the_gpu": round(tot_sort / S * 1e3, 3), "wall_of_the_whole_loop": round(tot_wall / S * 1e3, 3),
                                    "walker_lists_on_the_host": round(load_batches.walker_seconds * 1e3, 3), "record_starts_on_the_host(rb3h_strand_pairs)": round(load_batches.pairs_seconds * 1e3, 3),
                                    "note": "suffix sorting is excluded by the metric's definition (SURVEY 8(d): libsais in the reference); the walker lists are made on the device inside the timed merge calls since round 5 "
                                            "(walker_lists_on_the_host is 0 unless --host-walkers); where the records of a batch start (what the forward-strand upload needs) is found once before the timed steps "
                                            "here -- the CLI's reader knows it as it parses"},
        "h2d": {"symbols_per_step": int(sym_step), "bytes_over_pcie_per_step": int(sym_step // 2) if not args.full_upload else int(sym_step), "how": "rb3gpu_sorter_upload_fwd: forward strands copied, reverse complements written on the device" if not args.full_upload else "rb3gpu_sorter_upload: both strands copied", "GB/s_of_text": round(nsym / max(1e-9, tot_h2d) / 1e9, 2), "source": "pageable (staged)" if args.no_pinned else "page-locked (rb3gpu_pinned_alloc): one DMA per batch",
                "overlapped_with_the_merge_of_the_batch_before": bool(bl.overlap),
                "what_phases_ms_per_step.h2d_is": ("queueing the copies of batch i+1 (rb3gpu_sorter_upload_fwd_begin) before the merge of batch i is called + what the copy engine still needs once that merge has returned (rb3gpu_sorter_upload_end); the copies run beside the merge's kernels, as in the CLI (sorter thread) and the reference (reader, build.c:203-239)" if bl.overlap else "the upload call from start to completion, nothing beside it"),
                "ms_per_step_with_nothing_beside_it": None if h2d_serial is None else round(h2d_serial * 1e3, 3)},
        "roofline": chain_roofline(int(rows_launch), ms_chain, "text", load_pmc_traffic("k_chain_mtb152"),
                                   "k_chain<list,mixed,tent,text>: average over the %d launches of the timed steps (run-coded index, intervals of up to %d matching suffixes); `achieved` prices SURVEY 8(d)'s 208 B per LF step over the kernel's HIP-event time; "
                                   "traffic = FETCH_SIZE/WRITE_SIZE of the committed --pmc passes (profiles/r6_pmc_k_chain_mtb152.json); the index (<= %.0f MB) sits in L2 + Infinity Cache, so this kernel is bound by latency and instruction issue, not by HBM (aux_large_index is the HBM-resident case)" % (st["n_rank_launches"], K - 1, st["bytes_index"] / 1e6)),
        "roofline_path": {"bound": "hbm", "formula": "SURVEY 8(d): (217 B x symbols merged + bytes(B1 old) + bytes(B1 new) per round) / merge-path seconds / 8 TB/s", "algorithmic_bytes_per_step": int(path_bytes // S),
                          "achieved": round(path_bytes / dt / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(path_bytes / dt / 1e9 / HBM_PEAK_GBS, 5)},
        "roofline_rebuild": {"bound": "hbm", "kernel": "k_reb_group + k_place (+ window kernels on what they hand on)", "algorithmic_bytes_per_step": int(st["bytes_rebuild"] // S), "ms_per_step": round(st["ms_build"] / S, 3),
                             "achieved": round(st["bytes_rebuild"] / max(1e-9, st["ms_build"]) / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(st["bytes_rebuild"] / max(1e-9, st["ms_build"]) / 1e6 / HBM_PEAK_GBS, 5),
                             "note": "streaming roofline: 9 B per batch row + old block array + new block array per round"},
    }
    if refsig is not None:
        out["aux_mtb152_reference_signature"] = refsig
    if mrg_idx is not None:
        out["aux_merge_index"] = mrg_idx
    bl.close()
    if args.only == "headline":
        emit_json(out)
        return
    if not args.no_cpu_baseline:
        cb = reference_prefix(files, args.mtb_ref_prefix, K)
        if cb is None:   # the reference binary did not travel: the OpenMP port of the oracle on one round
            from ropebwt3_amd import host
            from tests import util
            orc = util.Oracle()
            b1 = host.build_bwt(texts[0])
            b2 = host.build_bwt(texts[1])
            t = time.time()
            orc.mg_rank(b1, b2, os.cpu_count() or 1)
            e = time.time() - t
            cb = {"value": round(b2.size / e / 1e9, 6), "unit": "Gbp/s", "cores": os.cpu_count() or 1, "kind": "port", "sample": "rank phase of round 1 (genome 1 into the index of genome 0) by oracle/liboracle.so"}
        out["cpu_baseline"] = cb
        if K >= args.mtb_3b_first + args.mtb_3b_batch and args.mtb_3b_batch > 0:
            cb3 = reference_many_chains(files, args.mtb_3b_first, args.mtb_3b_batch)
            if cb3 is not None:
                out["cpu_baseline_config3b"] = cb3
    if not args.no_aux:
        out["aux_cli_build"] = cli_build(files, K, gold)
        # everything around the metric in one place
        out["end_to_end"] = {"in_process_loop_wall_ms_per_build": out["not_counted_ms_per_step"]["wall_of_the_whole_loop"],
                             "of_which": {"merge_path_incl_h2d": out["ms_per_step"], "suffix_sorting_on_the_gpu": out["not_counted_ms_per_step"]["suffix_sorting_on_the_gpu"]},
                             "walker_lists_on_the_host_ms_per_build": out["not_counted_ms_per_step"]["walker_lists_on_the_host"],
                             "cli_build_wall_s": out["aux_cli_build"].get("build_wall_s"), "cli_command": out["aux_cli_build"].get("command"),
                             "cli_fmd_identical_to_reference": out["aux_cli_build"].get("fmd_identical_to_reference"),
                             "single_strand_input_Gbp_per_s_through_the_cli": round(K * L / 1e9 / out["aux_cli_build"]["build_wall_s"], 3) if out["aux_cli_build"].get("build_wall_s") else None,
                             "note": "the CLI reads 152 FASTA files, sorts every batch on the GPU (sorter thread), merges, packs the .fmd on the GPU and writes it: all overlapped; SURVEY 8(d) asks for this next to the metric"}
        try:
            out["aux_cfg2"] = cfg2_step(local_rank, 10, 3, MTB_L, args.div)
            out["aux_reads_regime"] = reads_regime(mk, args.aux_reads)
            if args.large_index > 0:
                out["aux_large_index"] = large_index_regime(mk, args.large_index, 1000000)
            if args.index_8g > 0:
                out["aux_index_8g"] = index_8g(args.index_8g, local_rank)
            if args.peer_rounds > 0:   # VERDICT r5 "next" 8: the lock-step rounds of the interval-sharded merge without the host, ranks = threads sharing this GPU
                from tools import probe_sh_peer
                rows = probe_sh_peer.measure([args.peer_rounds], worlds=(1, 2, 4), modes=(0, 1), index_log2=24, reps=2)
                out["aux_interval_peer_rounds"] = {"what": "rb3gpu_sh_merge of %d reads x 150 bp (both strands) into an index of 2^24 symbols cut into `world` intervals, ranks = threads with a handle each, ALL ON THIS ONE GPU: "
                                                           "us of the walk per lock-step round.  peer rounds: one kernel per rank and round that stores the next states straight into the owner's receive buffer, streams waiting "
                                                           "for each other's events (rb3gpu_comm_t.stream_barrier); driven by the host: read-back of the split sizes + all-gather + all-to-all per round (rounds 4-5).  "
                                                           "Never run on two devices; larger batches: profiles/r6_sh_peer_rounds.txt" % args.peer_rounds,
                                                   "rows": rows}
        except Exception as e:   # (a box with less free memory than a leg needs must not lose the headline)
            out["aux_error"] = repr(e)[:300]
        # BASELINE configs[3] / [4] in shape at 1/10 and at real contig sizes: minutes of box time each, so they are RECORDED runs (tools/r6/scale_reads.sh,
        # tools/r6/scale_hap.sh through the CLI, every 64th row of every merge LF-checked), not legs of this line; `--scale reads|hap` runs one now
        rec = {}
        for name, fn in (("aux_reads_18g", "r6_scale_reads_60m.json"), ("aux_haplotypes_4x3g", "r6_scale_hap_4x3g.json")):
            try:
                d = json.load(open(os.path.join(ROOT, "profiles", fn)))
                d.pop("samples_as_the_index_grows", None)
                rec[name] = dict(d, recorded_in="profiles/" + fn)
            except (OSError, ValueError):
                pass
        if rec:
            out["scale_runs_recorded"] = rec
    for f in files:
        try:
            os.unlink(f)
        except OSError:
            pass
    try:
        os.rmdir(tmp)
    except OSError:
        pass
    emit_json(out)


if __name__ == "__main__":
    main()
