#!/usr/bin/env python3
"""bench.py -- merge-path throughput of the MI355X engine (BASELINE.json metric:
"Gbp/sec indexed (build merge)").

A STEP is one pass of the hot path over one batch: rb3gpu_merge_plain_dev() = LF array of the
partial BWT B2 + all LF chains against the accumulated BWT B1 (rank) + interleave/rebuild of the
block array, with B1 and B2 already resident in HBM and the result discarded (commit=0) so that
every step does identical work.  Workload at N=1 = BASELINE.json configs[1] as SURVEY 8(d)
defines it without network access: G0 = 4.4 Mbp of uniform random ACGT (seed 1), G1 = G0 with
0.1 % substitutions (seed 2); the step merges G1 (both strands, 8,800,002 symbols, 2 strings)
into the index of G0.  At N>1 the batch grows with N (weak scaling): it holds N genomes G_1..G_N
(seeds 2..N+1, 8,800,002 symbols per GPU); the index is replicated, the batch's LF walkers are
sharded across the ranks by text range and pos[] is combined with one RCCL all-reduce(MAX) per step
(ropebwt3_amd/multi.py); value = symbols merged by the whole job / max-over-ranks time.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_SYMBOL_RANK = 208   # 16 B row entry r/w + 64 B directory line + 128 B block line per LF step
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench] " + msg, file=sys.stderr, flush=True)


def gen_genomes(n, rate, seed0, seeds):
    from tests import util
    g0 = util.random_genome(np.random.default_rng(seed0), n)
    return g0, [util.mutate(np.random.default_rng(s), g0, rate) for s in seeds]


def cpu_baseline(b1, b2, seconds_cap=60.0):
    """Time the merge of the SAME workload on the host cores: the unmodified reference
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
            "sample": "the full N=1 step (%d symbols, %d strings -> only %d of the %d threads have work, as in the reference's kt_for over strings)" % (b2.size, n_str, min(n_str, cores), cores)}


def load_pmc_traffic(workload):
    """HBM bytes per k_chain launch from the committed rocprofv3 --pmc passes (profiles/), if any."""
    fn = os.path.join(ROOT, "profiles", "r1_pmc_k_chain.json")
    try:
        d = json.load(open(fn))
        if d.get("workload") == workload:
            return d.get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--genome-len", type=int, default=4400000)
    ap.add_argument("--div", type=float, default=0.001)
    ap.add_argument("--split", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sharded", action="store_true", help="use the multi-GPU code path even with one rank")
    ap.add_argument("--walker-step", type=int, default=512)
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
    from ropebwt3_amd import Rb3Gpu, host

    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: torch.cuda.is_available() is False and the engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    t0 = time.time()
    sharded = world > 1 or args.sharded
    g0, gs = gen_genomes(args.genome_len, args.div, 1, [2 + i for i in range(world)])
    from tests import util
    b1 = host.build_bwt(util.make_text([g0]))
    walkers = None
    if sharded:
        b2, walkers = host.build_bwt_walkers(util.make_text(gs), args.walker_step)
    else:
        b2 = host.build_bwt(util.make_text(gs))
    log("inputs: B1 %d symbols, B2 %d symbols per GPU; host suffix sorting %.1f s (not timed)" % (b1.size, b2.size, time.time() - t0))

    h = Rb3Gpu(device=local_rank, split_log2=args.split, verbose=1)
    h.from_plain(b1)
    d_b2 = h.dev_upload(b2)

    def barrier():
        h.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    if sharded:
        from ropebwt3_amd import multi
        if world == 1 and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
        pos = torch.empty(b2.size, dtype=torch.int64, device="cuda")

        def step(commit=False):
            return multi.merge_sharded(h, d_b2, b2.size, walkers, args.walker_step, dist, rank, world, pos, commit=commit, sync=torch.cuda.synchronize)
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
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    st = h.stats()

    # one committed merge + export, to make sure the timed path produces the right index
    step(commit=True)
    acc = h.get_acc()
    assert acc[6] == b1.size + b2.size

    if rank == 0:
        n_sym = b2.size * args.steps
        value = n_sym / dt / 1e9
        ms_chain = st["ms_chain"] / max(1, st["n_rank_launches"])
        algo_bytes = ALGO_BYTES_PER_SYMBOL_RANK * b2.size // world   # rows recorded per launch on one rank
        achieved = algo_bytes / (ms_chain * 1e-3) / 1e9
        workload = "cfg2-synthetic-mtb1: merge %d genome(s) G_i = G0 + 0.1%% subs (%d bp each, both strands, %d symbols, %d strings) into the index of G0 (%d symbols)" % (world, args.genome_len, b2.size, 2 * world, b1.size)
        out = {
            "metric": "Gbp/s indexed (build merge)", "value": round(value, 6), "unit": "Gbp/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64", "data": "synthetic",
            "config": {"workload": workload, "symbols_per_step_per_gpu": int(b2.size // world), "strings_per_step": int((b2 == 0).sum()),
                       "index_symbols": int(b1.size),
                       "parallelism": ("replicated index, walkers sharded by text range over %d GPUs, all-reduce(MAX) of pos[] per step" % world) if sharded else "single GPU",
                       "split_log2": args.split, "lf_steps_per_step": int(st["n_lf_steps"] // max(1, args.steps))},
            "phases_ms_per_step": {"lf": round(st["ms_lf"] / args.steps, 4), "rank": round(st["ms_rank"] / args.steps, 4),
                                   "rebuild": round(st["ms_build"] / args.steps, 4)},
            "roofline": {"bound": "hbm", "kernel": "k_chain", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": load_pmc_traffic("cfg2"),
                         "algorithmic_bytes_per_launch": algo_bytes, "ms_per_launch": round(ms_chain, 4),
                         "note": "few-long-strings regime: the kernel is bound by dependent-load latency, not bandwidth"},
        }
        if world == 1 and not args.no_cpu_baseline and not args.sharded:
            out["cpu_baseline"] = cpu_baseline(b1, b2)
        print(json.dumps(out), flush=True)
    h.dev_free(d_b2)
    h.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
