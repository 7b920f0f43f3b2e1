"""Multi-GPU merge: index replicated on every GPU, the LF walkers of one batch sharded by text
range, one RCCL all-reduce(MAX) of pos[] per merge (SURVEY 8(e) option 2).

One process per GPU (torch.distributed, backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests).
The reference has no distributed code; what is sharded here is the kt_for over strings of
fm-index.c:217-224, generalised to walkers that start inside strings.

The orchestration only needs four engine calls (include/rb3gpu.h): mg_begin, mg_walk, the pos[]
buffer, mg_finish.  `engine` is an Rb3Gpu in production; the gloo tests pass a CPU stand-in with
the same methods so that the partition / hand-off / collective logic runs without a GPU.
"""
import numpy as np

WK_CHECK = 2
NSTEPS_INF = 1 << 60   # rb3h_build_bwt_walkers uses INT64_MAX/2 for "ends at the start of the string"


def partition(walkers, world, step):
    """Cut the text-ordered walker list into `world` contiguous slices of about equal work.
    Returns the world+1 slice bounds (slices may be empty when there are fewer walkers than ranks).
    Work of a walker ~ min(nsteps, step)."""
    w = np.minimum(walkers[:, 2], step).astype(np.float64)
    cum = np.concatenate([[0.0], np.cumsum(w)])
    bounds = [0]
    for r in range(1, world):
        b = int(np.searchsorted(cum, cum[-1] * r / world))
        bounds.append(min(max(b, bounds[-1]), len(walkers)))
    bounds.append(len(walkers))
    return bounds


def slice_plan(walkers, bounds, rank):
    """What rank `rank` does: its walkers, the row where the territory of the slice below begins
    (-1 if its lowest walker ends at the start of a string) and the rank it receives a hand-off
    value from (-1 if its top walker's segment starts a string end, i.e. nothing flows in)."""
    lo, hi = bounds[rank], bounds[rank + 1]
    mine = walkers[lo:hi]
    stop_row, src = -1, -1
    if hi > lo:
        if lo > 0 and mine[0, 2] < NSTEPS_INF:
            stop_row = int(walkers[lo - 1, 0])      # start row of the top walker of the slice below
        if hi < len(walkers) and walkers[hi, 2] < NSTEPS_INF:
            src = next(r for r in range(rank + 1, len(bounds) - 1) if bounds[r + 1] > bounds[r])  # owner of walkers[hi]
    return mine, stop_row, src


def merge_sharded(engine, d_bwt, length, walkers, step, dist, rank, world, pos, commit=True, sync=None):
    """One merge of a batch whose partial BWT (d_bwt, length) is resident on every rank.

    walkers: the full (n, 4) int64 walker list of the batch in text order (host array, identical
             on every rank).
    pos:     a torch int64 tensor of `length` elements on the engine's device; the engine records
             into it (rb3gpu_mg_begin d_pos_ext) and it is all-reduced across ranks.
    sync:    callable that makes the engine's work visible to torch collectives and vice versa
             (torch.cuda.synchronize on GPUs; a no-op for the CPU stand-in).
    Returns the number of hand-off rounds that were needed.
    """
    import torch
    sync = sync or (lambda: None)
    bounds = partition(walkers, world, step)
    mine, stop_row, src = slice_plan(walkers, bounds, rank)
    engine.mg_begin(d_bwt, length, pos.data_ptr())
    send = -1                                    # exact value some walker of mine arrived at stop_row with
    if len(mine):
        send = engine.mg_walk(mine, stop_row)
    pending = src >= 0                           # my top walker still waits for the value flowing in from above
    rounds = 0
    while True:
        sync()
        t = torch.tensor([send, 1 if pending else 0], dtype=torch.int64, device=pos.device)
        gathered = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        if not any(int(g[1]) for g in gathered):
            break
        rounds += 1
        if rounds > world + 1:
            raise RuntimeError("hand-off did not converge")   # the topmost pending boundary resolves every round
        val = int(gathered[src][0]) if pending else -1
        if pending and val >= 0:
            # the exact insertion point of my top walker's start row has arrived: one fix-up walker in
            # check mode; it stops at the first recorded row, or at stop_row, where its arrival value is
            # what the slice below is waiting for
            top = mine[-1]
            fix = np.array([[top[0], val, NSTEPS_INF, WK_CHECK]], dtype=np.int64)
            got = engine.mg_walk(fix, stop_row)
            pending = False
            if send < 0:
                send = got
    sync()
    dist.all_reduce(pos, op=dist.ReduceOp.MAX)   # unset rows are -1; every recorded value is exact
    sync()
    engine.mg_finish(commit)
    return rounds


def tree_merge(engine, dist, rank, world, device, sync=None):
    """Combine the per-rank indexes of a partitioned build into rank 0's index.

    Every rank has built the index of ITS contiguous slice of the input (slice r before slice r+1).
    Round k merges the index of rank r + 2^k into rank r for r = 0 mod 2^(k+1): the right-hand index
    is exported as a plain BWT on its GPU, sent over RCCL (xGMI) and merged with rb3gpu_merge_plain_dev.
    merge(A, B) ranks every sentinel of B after those of A (fm-index.c:147, 164), so merging adjacent
    slices left to right reproduces the BWT of the whole input in input order.
    Returns the number of symbols this rank holds afterwards (rank 0: everything).
    """
    import torch
    sync = sync or (lambda: None)
    stride = 1
    while stride < world:
        if rank % (2 * stride) == 0 and rank + stride < world:
            n = torch.zeros(1, dtype=torch.int64, device=device)
            dist.recv(n, src=rank + stride)
            buf = torch.empty(int(n.item()) + 16, dtype=torch.uint8, device=device)
            dist.recv(buf, src=rank + stride)
            sync()
            engine.merge_plain_dev(buf.data_ptr() if hasattr(buf, "data_ptr") else buf, int(n.item()), True)
        elif rank % (2 * stride) == stride:
            tot = engine.get_tot()
            buf = torch.empty(tot + 16, dtype=torch.uint8, device=device)
            engine.export_plain_dev(buf.data_ptr())
            sync()
            dist.send(torch.tensor([tot], dtype=torch.int64, device=device), dst=rank - stride)
            dist.send(buf, dst=rank - stride)
        stride *= 2
    return engine.get_tot()
