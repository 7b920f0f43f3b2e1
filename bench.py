#!/usr/bin/env python3
"""bench.py -- merge-path throughput of the MI355X engine (BASELINE.json metric:
"Gbp/sec indexed (build merge)").

A STEP is one pass of the hot path over one batch: LF array of the partial BWT B2 + all LF walkers
against the accumulated BWT B1 (rank) + interleave/rebuild of the block array, with B1, B2 and the
walker list already resident / in hand and the result discarded (commit=0) so that every step does
identical work.

N=1 workload = BASELINE.json configs[1] as SURVEY 8(d) defines it without network access:
G0 = 4.4 Mbp of uniform random ACGT (seed 1), G1 = G0 with 0.1 % substitutions (seed 2); the step
merges G1 (both strands, 8,800,002 symbols, 2 strings) into the index of G0.

N>1 (weak scaling, one process per GPU): the input is partitioned across the GPUs -- rank r merges
its own batch G_{r+1} (seed 2+r, 8,800,002 symbols) into the index its GPU holds; no data-path
collective inside a step.  value = symbols merged by all ranks / max-over-ranks time.  The
partitioned build ends with a binary tree of whole-index merges (ropebwt3_amd.multi.tree_merge, plain
BWTs over RCCL/xGMI); its time is reported separately as tree_merge_ms.  `--sharded` selects the
alternative decomposition (one batch of N genomes, walkers sharded by text range, all-reduce of pos[]).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_STEP = 208          # 16 B row entry r/w + 64 B directory line + 128 B block line per LF step
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


def load_pmc_traffic():
    """HBM bytes per k_chain launch from the committed rocprofv3 --pmc passes (profiles/), if any."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "r1_pmc_k_chain.json"))).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


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
    ach = ALGO_BYTES_PER_STEP * b2.size / (ms_chain * 1e-3) / 1e9
    return {"workload": "reads regime: merge %d x 150 bp reads (both strands, %d symbols, %d strings) into an index of %d symbols" % (n_reads, b2.size, 2 * n_reads, b1.size),
            "value": round(b2.size / dt / 1e9, 4), "unit": "Gbp/s", "ms_per_step": round(dt * 1e3, 3),
            "roofline": {"bound": "hbm", "kernel": "k_chain", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                         "ms_per_launch": round(ms_chain, 4), "lf_steps_per_s": round(b2.size / (ms_chain * 1e-3) / 1e9, 3)}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--genome-len", type=int, default=4400000)
    ap.add_argument("--div", type=float, default=0.001)
    ap.add_argument("--walker-step", type=int, default=384, help="text distance between LF walkers handed to the engine")
    ap.add_argument("--plain-abi", action="store_true", help="use rb3gpu_merge_plain_dev (the reference's signature, no walker list)")
    ap.add_argument("--sharded", action="store_true", help="N>1: one batch of N genomes, walkers sharded by text range + all-reduce")
    ap.add_argument("--row-words", action="store_true", help="walk row words (BWT + sampled inverse suffix array from the host sorter) instead of text-order words")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-aux", action="store_true", help="skip the auxiliary reads-regime measurement")
    ap.add_argument("--aux-reads", type=int, default=100000)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world

    import torch
    import torch.distributed as dist
    from ropebwt3_amd import Rb3Gpu, host, multi
    from tests import util

    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: torch.cuda.is_available() is False and the engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.sharded
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)

    t0 = time.time()
    sharded = args.sharded
    seeds = [2 + i for i in range(world)] if sharded else [2 + rank]
    g0, gs = gen_genomes(args.genome_len, args.div, 1, seeds)
    b1 = host.build_bwt(util.make_text([g0]))
    walkers = None
    text2 = util.make_text(gs)
    text_words = not (args.plain_abi or args.row_words or sharded)  # default: BWT + text-order words from the GPU suffix sorter
    if args.plain_abi and not sharded:
        b2 = host.build_bwt(text2.copy())
    elif text_words:
        b2 = host.build_bwt(text2.copy())
        walkers = host.walkers_text(text2, args.walker_step)
    else:
        b2, walkers = host.build_bwt_walkers(text2.copy(), args.walker_step)
    log("inputs: B1 %d symbols, B2 %d symbols on each GPU; host suffix sorting %.1f s (not timed)" % (b1.size, b2.size, time.time() - t0))

    h = Rb3Gpu(device=local_rank, verbose=1)
    h.from_plain(b1)
    d_tw = None
    if text_words:  # the batch as the GPU suffix sorter leaves it in HBM (sorting is outside the metric)
        d_b2, d_tw = h.sort_text(text2)
        assert np.array_equal(h.dev_download(d_b2, b2.size), b2), "GPU and host suffix sorters disagree"
    else:
        d_b2 = h.dev_upload(b2)

    def barrier():
        h.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    if sharded:
        pos = torch.empty(b2.size, dtype=torch.int64, device=dev)

        def step(commit=False):
            multi.merge_sharded(h, d_b2, b2.size, walkers, args.walker_step, dist, rank, world, pos, commit=commit, sync=torch.cuda.synchronize)
    elif text_words:
        def step(commit=False):
            h.merge_text_dev(d_b2, d_tw, b2.size, walkers, commit=commit)
    elif walkers is not None:
        def step(commit=False):
            h.merge_plain_dev_walkers(d_b2, b2.size, walkers, commit=commit)
    else:
        def step(commit=False):
            h.merge_plain_dev(d_b2, b2.size, commit=commit)

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
    if world > 1:
        dist.barrier()
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    st = h.stats()

    aux_t = None
    if rank == 0 and world == 1 and not args.no_aux and text_words:
        # the same step (same index, nothing committed yet) through the other two entry points, for the record:
        # what the batch has to come with matters
        _, w_rows = host.build_bwt_walkers(text2.copy(), args.walker_step)
        d_b2h = h.dev_upload(b2)

        def timed(fn, reps=10):
            fn()
            h.sync()
            t = time.perf_counter()
            for _ in range(reps):
                fn()
            h.sync()
            return (time.perf_counter() - t) / reps
        aux_t = (timed(lambda: h.merge_plain_dev_walkers(d_b2h, b2.size, w_rows, commit=False)), timed(lambda: h.merge_plain_dev(d_b2h, b2.size, commit=False)))
        h.dev_free(d_b2h)

    # one committed merge, to make sure the timed path produces a consistent index
    step(commit=True)
    acc = h.get_acc()
    assert acc[6] == b1.size + b2.size

    tree_ms = None
    if world > 1 and not sharded:  # the closing phase of a partitioned build: merge the per-GPU indexes into rank 0
        barrier()
        t = time.perf_counter()
        tot = multi.tree_merge(h, dist, rank, world, dev, sync=torch.cuda.synchronize)
        barrier()
        tree_ms = (time.perf_counter() - t) * 1e3
        if rank == 0:
            assert tot == world * (b1.size + b2.size)

    if rank == 0:
        sym_per_step = b2.size if sharded else b2.size * world
        value = sym_per_step * args.steps / dt / 1e9
        ms_chain = st["ms_chain"] / max(1, st["n_rank_launches"])
        rows_per_launch = b2.size // world if sharded else b2.size
        algo_bytes = ALGO_BYTES_PER_STEP * rows_per_launch
        achieved = algo_bytes / (ms_chain * 1e-3) / 1e9
        if sharded:
            par = "one batch of %d genomes; index replicated, walkers sharded by text range over %d GPUs, all-reduce(MAX) of pos[] per step" % (world, world)
        elif world > 1:
            par = "input partitioned over %d GPUs: every GPU merges its own batch into the index it holds; no collective inside a step; closing tree merge reported as tree_merge_ms" % world
        else:
            par = "single GPU"
        out = {
            "metric": "Gbp/s indexed (build merge)", "value": round(value, 6), "unit": "Gbp/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64", "data": "synthetic",
            "config": {"workload": "cfg2-synthetic-mtb1: per GPU, merge G_i = G0 + 0.1%% substitutions (%d bp, both strands, %d symbols, 2 strings) into the index of G0 (%d symbols)" % (args.genome_len, rows_per_launch, b1.size),
                       "symbols_per_step_per_gpu": int(rows_per_launch), "index_symbols": int(b1.size), "parallelism": par,
                       "entry_point": "rb3gpu_merge_plain_dev (reference signature, SA-order walkers)" if walkers is None else
                                      "rb3gpu_merge_text_dev (BWT + text-order words = inverse suffix array, both from the GPU suffix sorter; walkers by text position, step %d, %d walkers)" % (args.walker_step, len(walkers)) if text_words else
                                      "rb3gpu_merge_plain_dev_walkers (BWT + sampled inverse suffix array from the host suffix sorter, text step %d, %d walkers)" % (args.walker_step, len(walkers)),
                       "lf_steps_per_step": int(st["n_lf_steps"] // max(1, args.steps)), "rank_phase_fallbacks": int(st["n_fallbacks"])},
            "phases_ms_per_step": {"lf": round(st["ms_lf"] / args.steps, 4), "rank": round(st["ms_rank"] / args.steps, 4),
                                   "rebuild": round(st["ms_build"] / args.steps, 4)},
            "roofline": {"bound": "hbm", "kernel": "k_chain", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": load_pmc_traffic(),
                         "algorithmic_bytes_per_launch": algo_bytes, "ms_per_launch": round(ms_chain, 4),
                         "note": "random-access bound (per LF step one 128-B block line read and one 8-B record written at random rows, ~4.6 TB/s of 64-B sectors; the batch side is streamed; waves wait on memory 75% of their cycles, profiles/r1_pmc_sq.txt); aux_reads_regime is the same kernel on 200 k short strings"},
        }
        if tree_ms is not None:
            out["tree_merge_ms"] = round(tree_ms, 3)
        if world == 1 and not args.no_aux:
            out["aux_reads_regime"] = reads_regime(lambda: Rb3Gpu(device=local_rank, verbose=1), args.aux_reads)
        if aux_t is not None:
            t_rows, t_abi = aux_t
            out["aux_entry_points"] = {
                "note": "same workload, same handle; `value` above is the first line of this table",
                "rb3gpu_merge_text_dev (BWT + inverse suffix array of the batch, as the GPU sorter leaves them; producing the words costs the sorter one 19-us kernel)": {"ms_per_step": round(dt / args.steps * 1e3, 4), "Gbp/s": round(value, 3)},
                "rb3gpu_merge_plain_dev_walkers (BWT + inverse suffix array sampled every %d positions, from the host sorter; the LF array of the batch is built inside the step)" % args.walker_step: {"ms_per_step": round(t_rows * 1e3, 4), "Gbp/s": round(b2.size / t_rows / 1e9, 3)},
                "rb3gpu_merge_plain_dev (the reference's signature: len + BWT only)": {"ms_per_step": round(t_abi * 1e3, 4), "Gbp/s": round(b2.size / t_abi / 1e9, 3)}}
        if world == 1 and not args.no_cpu_baseline and not sharded:
            out["cpu_baseline"] = cpu_baseline(b1, b2)
        print(json.dumps(out), flush=True)
    h.dev_free(d_b2)
    if d_tw is not None:
        h.dev_free(d_tw)
    h.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
