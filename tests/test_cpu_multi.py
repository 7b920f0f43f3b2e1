"""world_size-2/3 gloo tests of the multi-GPU orchestration (ropebwt3_amd/multi.py): walker
partition by text range, stop_row hand-off, fix-up round, all-reduce(MAX) of pos[].  The engine
calls are served by tests/fake_engine.py (numpy restatement of the walker kernel), the expected
pos[] comes from the CPU oracle."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from ropebwt3_amd import host, multi
    from tests import util
    from tests.fake_engine import FakeEngine
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(5)
        g0 = util.random_genome(rng, 6000)
        if case == "one_long_string":          # a single string cut across every rank boundary
            seqs, step, rev = [util.mutate(rng, g0, 0.01)], 200, False
        elif case == "duplicate":              # identical to indexed text: nothing converges, pure hand-off chain
            seqs, step, rev = [g0[:3000].copy()], 150, False
        elif case == "reads":                  # many short strings: no walker inside strings at all
            seqs, step, rev = util.reads_from(rng, g0, 40, 60), 1024, True
        else:                                  # mixed
            seqs, step, rev = [util.mutate(rng, g0, 0.02)[:2500], g0[100:130].copy(), util.mutate(rng, g0, 0.005)[1000:5000]], 128, True
        orc = util.Oracle()
        b1 = orc.bwt(util.make_text([g0]))
        t2 = util.make_text(seqs, rev=rev)
        b2, walkers = host.build_bwt_walkers(t2, step)
        rb, _ = orc.mg_rank(b1, b2)
        eng = FakeEngine(b1)
        pos = torch.full((b2.size,), -1, dtype=torch.int64)
        real_begin = eng.mg_begin
        eng.mg_begin = lambda d, n, ptr: real_begin(d, n, ptr, pos_tensor=pos)
        rounds = multi.merge_sharded(eng, b2, b2.size, walkers, step, dist, rank, world, pos)
        ok = bool(np.array_equal(pos.numpy(), rb >> 6))
        bounds = multi.partition(walkers, world, step)
        q.put((rank, ok, rounds, eng.steps, bounds[rank + 1] - bounds[rank]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,case", [(2, "one_long_string"), (3, "one_long_string"), (2, "duplicate"), (3, "duplicate"), (2, "reads"), (2, "mixed"), (3, "mixed")])
def test_sharded_merge_gloo(world, case):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert all(r[1] for r in res), res
    if case == "one_long_string":
        assert all(r[4] > 0 for r in res)              # every rank really had walkers
        assert max(r[2] for r in res) >= 1             # and at least one hand-off round happened
    if case == "duplicate":
        assert max(r[2] for r in res) == world - 1     # the value has to travel down rank by rank
    if case == "reads":
        assert max(r[2] for r in res) == 0             # strings are not cut: no hand-off at all


def test_partition_and_plan():
    from ropebwt3_amd import multi
    INF = multi.NSTEPS_INF
    # two strings: the first with 3 checkpoints + sentinel, the second only a sentinel walker
    w = np.array([[10, -1, INF, 0], [11, -1, 100, 0], [12, -1, 100, 0], [0, -2, 50, 0], [1, -2, INF, 0]], dtype=np.int64)
    b = multi.partition(w, 2, 100)
    assert b[0] == 0 and b[-1] == 5 and 0 < b[1] < 5
    mine0, stop0, src0 = multi.slice_plan(w, b, 0)
    mine1, stop1, src1 = multi.slice_plan(w, b, 1)
    assert stop0 == -1                                  # the lowest slice never hands down
    assert (src0 == 1) == (w[b[1], 2] < INF)            # rank 0 receives iff rank 1's lowest walker flows into it
    assert (stop1 == w[b[1] - 1, 0]) == (w[b[1], 2] < INF)
    # more ranks than walkers: empty slices are skipped when looking for the sender
    b = multi.partition(w[:2], 4, 100)
    plans = [multi.slice_plan(w[:2], b, r) for r in range(4)]
    owners = [r for r in range(4) if len(plans[r][0])]
    assert sum(len(p[0]) for p in plans) == 2
    if len(owners) == 2:
        assert plans[owners[0]][2] == owners[1]


def _tree_worker(rank, world, port, q, host_link=False):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from ropebwt3_amd import multi
    from tests import util
    from tests.fake_engine import FakePlainEngine
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(9)
        g0 = util.random_genome(rng, 1500)
        seqs = [util.mutate(rng, g0, 0.01) for _ in range(2 * world + 1)] + [g0[:40].copy(), g0[:40].copy()]
        cut = [len(seqs) * r // world for r in range(world + 1)]
        orc = util.Oracle()
        eng = FakePlainEngine(orc, orc.bwt(util.make_text(seqs[cut[rank]:cut[rank + 1]])))
        link = multi.TreeLink(dist, torch.device("cpu"), False) if host_link else None   # (what bench.py uses when ranks share a GPU)
        multi.tree_merge(eng, dist, rank, world, torch.device("cpu"), link=link)
        ok = True
        if rank == 0:
            ok = bool(np.array_equal(eng.b, orc.bwt(util.make_text(seqs))))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,host_link", [(2, False), (3, False), (4, False), (3, True)])
def test_tree_merge_gloo(world, host_link):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tree_worker, args=(r, world, port, q, host_link)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res


def _interval_worker(rank, world, port, case, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from ropebwt3_amd import host, multi
    from tests import util
    from tests.fake_engine import FakeShardEngine
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(17)
        g0 = util.random_genome(rng, 3000)
        orc = util.Oracle()
        cur = orc.bwt(util.make_text([g0] + util.reads_from(rng, g0, 10, 50)))
        bounds = multi.interval_bounds(cur.size, world)
        eng = FakeShardEngine(cur[bounds[rank]:bounds[rank + 1]])
        comm = multi.TorchComm(dist, rank, world, torch.device("cpu"), as_numpy=True)
        rounds = []
        for b in range(3):     # three batches in a row: the interval bounds move with every merge
            if case == "reads":
                seqs = util.reads_from(rng, g0, 30, 40, err=0.02)
            elif case == "dups":   # exact duplicates of indexed text and of each other: ties are broken by the sentinel order
                seqs = [g0[100:160].copy(), g0[100:160].copy(), g0[:30].copy()] + util.reads_from(rng, g0, 5, 40)
            else:                  # strings of very different lengths
                seqs = [util.mutate(rng, g0, 0.01)[:400], g0[5:9].copy(), util.mutate(rng, g0, 0.02)[1000:1100]]
            t2 = util.make_text(seqs, rev=(b != 1))
            b2 = host.build_bwt(t2.copy())
            # text-order words of the batch from its suffix array (what rb3gpu_sort_text leaves in HBM)
            sa = np.array(sorted(range(t2.size), key=lambda i: _suffix_key(t2, i)), dtype=np.int64)
            isa = np.empty(t2.size, dtype=np.int64)
            isa[sa] = np.arange(t2.size)
            prev = np.concatenate([[0], t2[:-1]]).astype(np.int64)
            tw = (isa << 3) | prev
            assert np.array_equal(b2, prev[sa].astype(np.uint8))
            st = {}
            bounds = multi.merge_interval(eng, comm, bounds, b2, tw, t2.size, np.flatnonzero(t2 == 0), commit=True, stats=st)
            rounds.append(st["rounds"])
            cur = orc.merge(cur, b2)
            assert bounds[-1] == cur.size
            assert np.array_equal(eng.b, cur[bounds[rank]:bounds[rank + 1]]), (case, b, rank)
        q.put((rank, True, rounds))
    finally:
        dist.destroy_process_group()


def _suffix_key(t, i):
    """suffix of the multi-string text starting at i, cut after its sentinel, with the sentinel ranked by its position
    (the i-th sentinel sorts before the (i+1)-th, sais-ss.c:16-21)"""
    j = i
    while t[j] != 0:
        j += 1
    return tuple(int(x) + 1 for x in t[i:j]) + (0, j)


@pytest.mark.parametrize("world,case", [(2, "reads"), (3, "reads"), (2, "dups"), (2, "ragged"), (4, "ragged")])
def test_interval_sharded_merge_gloo(world, case):
    """merge_interval (north_star: index cut into intervals, one all-to-all per symbol) over gloo with a numpy stand-in per
    rank: after every merge the concatenation of the intervals is the oracle's merged BWT"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_interval_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res
    assert all(r[2] == res[0][2] for r in res)      # every rank went through the same number of rounds
    assert min(res[0][2]) >= 5                       # (longest string + 1)
