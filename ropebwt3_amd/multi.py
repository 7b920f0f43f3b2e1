"""Multi-GPU merge.  Three decompositions:

* merge_interval (north_star): the accumulated BWT is cut into contiguous INTERVALS of positions, one per
  GPU; the chains of a batch of short strings advance in lock step, every chain state is processed on the GPU whose
  interval contains its insertion point, and one all-to-all per symbol routes the new states to their owners
  (rb3gpu_sh_step / rb3gpu_sh_finish); the rebuild of every interval is local.
* merge_sharded (SURVEY 8(e) option 2, long strings): index replicated on every GPU, the LF walkers of one batch sharded
  by text range, one RCCL all-reduce(MAX) of pos[] per merge.
* tree_merge: partitioned input, one index per GPU, combined by a binary tree of whole-index merges.

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


class TreeLink:
    """point-to-point transport of a plain BWT between two ranks of the tree merge.  backend "nccl" (= RCCL over xGMI): device
    tensors, send/recv straight between the HBMs.  backend "gloo" (CPU tests, or several ranks sharing ONE GPU, which RCCL
    refuses): the BWT goes through host memory."""

    def __init__(self, dist, device, on_device, sync=None):
        import torch
        self.t, self.dist, self.device, self.on_device = torch, dist, device, on_device
        self.sync = sync or (lambda: None)

    def send_index(self, engine, dst):
        t = self.t
        tot = engine.get_tot()
        if self.on_device:
            buf = t.empty(tot + 16, dtype=t.uint8, device=self.device)
            engine.export_plain_dev(buf.data_ptr())
            self.sync()
            self.dist.send(t.tensor([tot], dtype=t.int64, device=self.device), dst=dst)
            self.dist.send(buf, dst=dst)
        else:
            b = engine.export_plain()
            self.dist.send(t.tensor([tot], dtype=t.int64), dst=dst)
            self.dist.send(t.from_numpy(b), dst=dst)
        return tot

    def recv_and_merge(self, engine, src):
        t = self.t
        if self.on_device:
            n = t.zeros(1, dtype=t.int64, device=self.device)
            self.dist.recv(n, src=src)
            buf = t.empty(int(n.item()) + 16, dtype=t.uint8, device=self.device)
            self.dist.recv(buf, src=src)
            self.sync()
            engine.merge_plain_dev(buf.data_ptr(), int(n.item()), True)
        else:
            n = t.zeros(1, dtype=t.int64)
            self.dist.recv(n, src=src)
            buf = t.empty(int(n.item()), dtype=t.uint8)
            self.dist.recv(buf, src=src)
            if hasattr(engine, "merge_plain_host"):
                engine.merge_plain_host(buf.numpy())          # (CPU stand-in of the tests)
            else:
                d = engine.dev_upload(buf.numpy())
                try:
                    engine.merge_plain_dev(d, int(n.item()), True)
                finally:
                    engine.dev_free(d)
        return int(n.item())


def tree_merge(engine, dist, rank, world, device, sync=None, link=None):
    """Combine the per-rank indexes of a partitioned build into rank 0's index.

    Every rank has built the index of ITS contiguous slice of the input (slice r before slice r+1).
    Round k merges the index of rank r + 2^k into rank r for r = 0 mod 2^(k+1): the right-hand index
    is exported as a plain BWT on its GPU, sent over RCCL (xGMI) and merged with rb3gpu_merge_plain_dev
    (rb3_fmi_merge, fm-index.c:251-277, with the right operand taken as its BWT).
    merge(A, B) ranks every sentinel of B after those of A (fm-index.c:147, 164), so merging adjacent
    slices left to right reproduces the BWT of the whole input in input order.
    Returns the number of symbols this rank holds afterwards (rank 0: everything).
    """
    link = link or TreeLink(dist, device, True, sync)
    stride = 1
    while stride < world:
        if rank % (2 * stride) == 0 and rank + stride < world:
            link.recv_and_merge(engine, rank + stride)
        elif rank % (2 * stride) == stride:
            link.send_index(engine, rank - stride)
        stride *= 2
    return engine.get_tot()



# ---------------------------------------------------------------------------------------------
# interval-sharded index (north_star): lock-step chains, all-to-all per symbol
# ---------------------------------------------------------------------------------------------

class TorchComm:
    """collectives of one rank over torch.distributed ("nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU tests).
    Buffers are torch tensors; `as_numpy`: the engine is a CPU stand-in that takes numpy views instead of device pointers.
    sync(): make the engine's stream and torch's stream see each other's work (torch.cuda.synchronize on GPUs)."""

    def __init__(self, dist, rank, world, device, sync=None, as_numpy=False):
        import torch
        self.t, self.dist, self.rank, self.world, self.device = torch, dist, rank, world, device
        self.sync = sync or (lambda: None)
        self.as_numpy = as_numpy

    def new_states(self, n):
        return self.t.empty((max(int(n), 1), 2), dtype=self.t.int64, device=self.device)

    def new_i64(self, n, fill):
        return self.t.full((max(int(n), 1),), fill, dtype=self.t.int64, device=self.device)

    def ptr(self, buf):
        return buf.numpy() if self.as_numpy else buf.data_ptr()

    def upload_states(self, buf, arr):
        if len(arr):
            buf[:len(arr)].copy_(self.t.from_numpy(np.ascontiguousarray(arr, dtype=np.int64)))
        self.sync()

    def free(self, buf):
        pass

    def all_gather(self, vec):
        v = self.t.tensor(np.asarray(vec, dtype=np.int64), device=self.device)
        if getattr(self, "_flat_ok", True):   # one collective into one tensor, one copy to the host
            try:
                out = self.t.empty((self.world * v.numel(),), dtype=v.dtype, device=self.device)
                self.dist.all_gather_into_tensor(out, v)
                return out.cpu().numpy().reshape(self.world, -1)
            except (RuntimeError, AttributeError, NotImplementedError):
                self._flat_ok = False
        out = [self.t.empty_like(v) for _ in range(self.world)]
        self.dist.all_gather(out, v)
        return np.stack([o.cpu().numpy() for o in out])

    def exchange(self, send, send_counts, recv_counts, recv):
        """all-to-all(v) of states: send is grouped by destination; returns the number of states received into recv"""
        n_in = int(np.sum(recv_counts))
        self.sync()
        if self.world == 1:
            if n_in:
                recv[:n_in].copy_(send[:n_in])
        else:
            self.dist.all_to_all_single(recv[:n_in], send[:int(np.sum(send_counts))], [int(x) for x in recv_counts], [int(x) for x in send_counts])
        self.sync()
        return n_in


class ThreadComm:
    """W ranks as W threads of ONE process on ONE GPU, each with its own engine handle (own HIP stream): the collectives are
    barriers plus device-to-device copies.  Drives the real engine through the sharded protocol without a multi-GPU node."""

    class Shared:
        def __init__(self, world):
            import threading
            self.world = world
            self.barrier = threading.Barrier(world)
            self.slots = [None] * world

    def __init__(self, shared, rank, engine):
        self.sh, self.rank, self.world, self.engine = shared, rank, shared.world, engine

    def new_states(self, n):
        return self.engine.dev_alloc(max(int(n), 1) * 16)

    def new_i64(self, n, fill):
        p = self.engine.dev_alloc(max(int(n), 1) * 8)
        self.engine.dev_memset(p, 0xff if fill == -1 else 0, max(int(n), 1) * 8)
        return p

    def ptr(self, buf):
        return buf

    def upload_states(self, buf, arr):
        if len(arr):
            self.engine.dev_upload_to(buf, np.ascontiguousarray(arr, dtype=np.int64))

    def free(self, buf):
        self.engine.dev_free(buf)

    def all_gather(self, vec):
        self.sh.slots[self.rank] = np.asarray(vec, dtype=np.int64).copy()
        self.sh.barrier.wait()
        out = np.stack(self.sh.slots)
        self.sh.barrier.wait()
        return out

    def exchange(self, send, send_counts, recv_counts, recv):
        off = np.concatenate([[0], np.cumsum(send_counts)]).astype(np.int64)
        self.sh.slots[self.rank] = (send, off)
        self.sh.barrier.wait()
        n_in = int(np.sum(recv_counts))
        at = 0
        for src in range(self.world):
            p, o = self.sh.slots[src]
            n = int(o[self.rank + 1] - o[self.rank])
            if n:
                self.engine.dev_copy(recv + at * 16, p + int(o[self.rank]) * 16, n * 16)
            at += n
        self.sh.barrier.wait()
        return n_in


def interval_bounds(n, world):
    """world contiguous intervals of about equal length over n positions.  Every rank must own at least one position (a handle
    without a block array cannot take part in rb3gpu_sh_step): an index smaller than the number of GPUs is not sharded."""
    if n < world:
        raise ValueError("an index of %d symbols cannot be cut into %d non-empty intervals: use fewer GPUs" % (n, world))
    return np.array([n * r // world for r in range(world + 1)], dtype=np.int64)


def merge_interval(engine, comm, bounds, d_bwt, d_tw, n2, sent_tp, commit=True, stats=None):
    """One merge of a batch of (short) strings into the interval-sharded index (north_star).

    engine:  this rank's handle; it holds the block array of interval comm.rank = [bounds[r], bounds[r+1]).
    d_bwt, d_tw: the batch's BWT and text-order words, replicated on every rank (device pointers of this rank).
    sent_tp: text positions of the sentinels of the batch (host int64 array, the same on every rank): where the chains start.
    Returns the interval bounds after the merge (every rank computes the same array).
    """
    rank, world = comm.rank, comm.world
    bounds = np.asarray(bounds, dtype=np.int64)
    # symbol totals of every interval -> C array of the whole BWT and this interval's additive offsets (mrope.c:76-88)
    acc = np.asarray(engine.get_acc(), dtype=np.int64)
    allc = comm.all_gather(np.diff(acc)[:6])                       # world x 6
    tot = allc.sum(axis=0)
    C = np.concatenate([[0], np.cumsum(tot)])[:6]
    pre = allc[:rank].sum(axis=0) if rank > 0 else np.zeros(6, dtype=np.int64)
    adj = C + pre - acc[:6]
    m1 = int(tot[0])                                               # every chain starts at ka = #sentinels of the index (fm-index.c:164)
    owner0 = int(np.sum(bounds[1:world] <= m1))
    d_ka = comm.new_i64(n2, -1)
    # two state buffers for the whole merge: a rank never holds more states than there are strings
    cap = len(sent_tp)
    cur = nxt = None
    try:
        cur, nxt = comm.new_states(cap), comm.new_states(cap)
        n_cur = cap if rank == owner0 else 0
        if n_cur:
            st = np.empty((n_cur, 2), dtype=np.int64)
            st[:, 0], st[:, 1] = sent_tp, m1
            comm.upload_states(cur, st)
        rows_here, rounds = 0, 0
        while True:
            rows_here += n_cur
            counts = engine.sh_step(n_cur, comm.ptr(cur), d_tw, comm.ptr(d_ka), adj, bounds, rank, comm.ptr(nxt))   # nxt: grouped by destination
            M = comm.all_gather(counts[:world])                        # M[s][d]: states rank s sends to rank d
            rounds += 1
            if int(M.sum()) == 0:
                break
            n_cur = comm.exchange(nxt, M[rank], M[:, rank], cur)
        R = comm.all_gather([rows_here])[:, 0]
        jlo = int(R[:rank].sum())
        if int(R.sum()) != n2:
            raise RuntimeError("sharded merge recorded %d of %d rows" % (int(R.sum()), n2))
        # (rb3gpu_sh_finish validates what this protocol can get wrong -- every row of the interval recorded by this rank, in
        # order, inside the interval -- with k_sh_localpos + k_pos_check; the sampled LF check of the single-GPU merge guards its
        # SPECULATIVE records, and there are none here: every state carries an exact insertion point)
        engine.sh_finish(jlo, rows_here, d_bwt, comm.ptr(d_ka), int(bounds[rank]), commit)
    finally:   # (an engine error in mid-loop must not leak the device buffers of a long-running build)
        for b in (cur, nxt, d_ka):
            if b is not None:
                comm.free(b)
    if stats is not None:
        stats["rounds"] = rounds
        stats["rows_per_rank"] = [int(x) for x in R]
    grow = np.concatenate([[0], np.cumsum(R)])
    return bounds + grow if commit else bounds


class _RawDev:
    """a device buffer the engine owns, seen by torch (no copy): __cuda_array_interface__ over the raw pointer"""

    def __init__(self, ptr, n_i64):
        self.__cuda_array_interface__ = {"shape": (int(n_i64),), "typestr": "<i8", "data": (int(ptr), False), "version": 2}


def sh_comm(h, dist, rank, world, dev, driver="rccl"):
    """the communicator rb3gpu_sh_merge runs over in a one-process-per-GPU job (bench.py --gpus N --mode interval).
    driver "rccl": the library's own (rb3gpu_rccl_*: grouped ncclSend/ncclRecv on the engine's stream; the 128-byte id travels
    through torch.distributed); "torch": two callbacks over torch.distributed (all_gather of the split sizes, all_to_all of views
    of the engine's send regions) -- also what "rccl" falls back to when librccl cannot be loaded.  Returns (comm, label)."""
    import os
    import torch
    from ropebwt3_amd import RcclComm, CallbackComm, Rb3GpuError, ipc_peer_enable

    def with_peer_rounds(comm, label, default_on=False):
        """the ranks are processes of ONE node: the lock-step rounds as peer rounds through HIP IPC where every rank can (collective).  On by default where the
        ranks share a GPU (test mode: run on hardware, tests/test_gpu_engine.py); between DISTINCT devices it has never run -- a mapping that does not come back
        would cost the leg its number --, so there it waits for RB3_IPC_PEER=1 (RB3_NO_IPC_PEER=1: never)"""
        on = (default_on or os.environ.get("RB3_IPC_PEER") == "1") and not os.environ.get("RB3_NO_IPC_PEER")
        if world > 1 and on:
            try:
                if ipc_peer_enable(h, comm):
                    label += " + PEER ROUNDS between the processes (rb3gpu_ipc_peer_enable: HIP IPC memory and event handles, a barrier in shared memory; the collectives above only carry the handles and the final exchange)"
            except Rb3GpuError as e:
                label += " [rb3gpu_ipc_peer_enable failed: %r]" % (e,)
        return comm, label

    why = ""
    if driver == "rccl":
        try:
            box = [RcclComm.unique_id() if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(box, src=0)
            return with_peer_rounds(RcclComm(h, rank, world, box[0]), "rb3gpu_rccl (ncclSend/ncclRecv grouped per round, on the engine's stream)")
        except (Rb3GpuError, OSError) as e:   # (every rank fails alike: the library is the same on all of them)
            why = " [rb3gpu_rccl unavailable: %r]" % (e,)

    if driver == "gloo":   # ranks that share a GPU (test mode): CPU tensors, the states staged through host memory
        def all_gather_g(vec):
            out = [torch.zeros(len(vec), dtype=torch.int64) for _ in range(world)]
            dist.all_gather(out, torch.from_numpy(np.ascontiguousarray(vec, dtype=np.int64)))
            return torch.stack(out).numpy()

        def exchange_g(d_send, stride, send_cnt, d_recv, recv_cnt):
            parts = [torch.from_numpy(h.dev_download_i64(d_send + d * stride * 16, int(send_cnt[d]) * 2)) if send_cnt[d] else torch.zeros(0, dtype=torch.int64) for d in range(world)]
            recv = [torch.zeros(int(recv_cnt[s_]) * 2, dtype=torch.int64) for s_ in range(world)]
            for peer in range(world):   # pairwise, the lower rank sends first
                if peer == rank:
                    recv[rank].copy_(parts[rank])
                elif rank < peer:
                    if parts[peer].numel(): dist.send(parts[peer], peer)
                    if recv[peer].numel(): dist.recv(recv[peer], peer)
                else:
                    if recv[peer].numel(): dist.recv(recv[peer], peer)
                    if parts[peer].numel(): dist.send(parts[peer], peer)
            got = torch.cat(recv).numpy()
            if got.size:
                h.dev_upload_to(d_recv, got)

        return with_peer_rounds(CallbackComm(rank, world, all_gather_g, exchange_g), "callbacks over gloo through host memory (ranks share a GPU: test mode)", default_on=True)

    def all_gather(vec):
        v = torch.as_tensor(np.ascontiguousarray(vec, dtype=np.int64), device=dev)
        out = torch.empty(world * v.numel(), dtype=torch.int64, device=dev)
        if world > 1:
            dist.all_gather_into_tensor(out, v)
        else:
            out.copy_(v)
        return out.cpu().numpy().reshape(world, -1)

    def exchange(d_send, stride, send_cnt, d_recv, recv_cnt):
        h.sync()
        ins = [torch.as_tensor(_RawDev(d_send + d * stride * 16, int(send_cnt[d]) * 2), device=dev) if send_cnt[d] else torch.empty(0, dtype=torch.int64, device=dev) for d in range(world)]
        outs, at = [], 0
        for s_ in range(world):
            n = int(recv_cnt[s_])
            outs.append(torch.as_tensor(_RawDev(d_recv + at * 16, n * 2), device=dev) if n else torch.empty(0, dtype=torch.int64, device=dev))
            at += n
        if world > 1:
            dist.all_to_all(outs, ins)
        elif ins[0].numel():
            outs[0].copy_(ins[0])
        torch.cuda.synchronize()

    return with_peer_rounds(CallbackComm(rank, world, all_gather, exchange), "callbacks over torch.distributed (all_gather_into_tensor + all_to_all of views of the engine's send regions)" + why)


def merge_interval_text(engine, comm, bounds, d_tprev, d_tw_slice, n2, sent_tp, commit=True, stats=None):
    """merge_interval_c with the BATCH sharded as well (rb3gpu_sh_merge_text): d_tprev = the symbol before every text position (1 byte each),
    d_tw_slice = the text-order words of this rank's text range only"""
    nb, rounds = engine.sh_merge_text(comm, bounds, d_tprev, d_tw_slice, n2, sent_tp, commit)
    if stats is not None:
        stats["rounds"] = rounds
        stats["rows_per_rank"] = [int(x) for x in np.diff(nb) - np.diff(np.asarray(bounds, dtype=np.int64))] if commit else None
    return nb


def merge_interval_c(engine, comm, bounds, d_bwt, d_tw, n2, sent_tp, commit=True, stats=None):
    """merge_interval with the lock-step loop inside the library (rb3gpu_sh_merge): same arguments, same result"""
    nb, rounds = engine.sh_merge(comm, bounds, d_bwt, d_tw, n2, sent_tp, commit)
    if stats is not None:
        stats["rounds"] = rounds
        stats["rows_per_rank"] = [int(x) for x in np.diff(nb) - np.diff(np.asarray(bounds, dtype=np.int64))] if commit else None
    return nb


class SoloComm(TorchComm):
    """the interval-sharded protocol with a single interval: no process group is touched"""

    def __init__(self, device, sync):
        import torch
        self.t, self.dist, self.rank, self.world, self.device = torch, None, 0, 1, device
        self.sync = sync
        self.as_numpy = False

    def all_gather(self, vec):
        return np.asarray(vec, dtype=np.int64)[None, :].copy()


def _reads_text(g, st, rng, read_len=150, err=0.01):
    """the batch text of reads g[s : s + read_len] (s in st) with substitution errors, every read followed by its reverse complement, each with its
    sentinel (io.c:84-102) -- tests.util.make_text on a list of reads, without a Python loop over millions of them"""
    r = np.lib.stride_tricks.sliding_window_view(g, read_len)[st]   # (a copy: fancy indexing of the view)
    k = int(rng.binomial(r.size, err))            # (the errors drawn as positions, not as a mask over every symbol: seconds per million reads)
    r.reshape(-1)[rng.integers(0, r.size, size=k)] = rng.integers(1, 5, size=k, dtype=np.uint8)
    n = r.shape[0]
    out = np.zeros((n, 2, read_len + 1), dtype=np.uint8)
    out[:, 0, :read_len] = r
    out[:, 1, :read_len] = np.array([0, 4, 3, 2, 1, 5], dtype=np.uint8)[r[:, ::-1]]   # complement of the reversed read
    return out.reshape(-1)


def _solo_interval_reference(reads_per_gpu, args, dev, local_rank):
    """bench_main's interval workload at world size 1 with the per-GPU sizes of the multi-GPU run: index of 2^26 symbols in one
    interval, reads_per_gpu reads per step"""
    import time
    import torch
    from ropebwt3_amd import Rb3Gpu
    from tests import util
    h1 = Rb3Gpu(device=local_rank, verbose=1)
    try:
        rng = np.random.default_rng(31)
        g = util.random_genome(rng, (1 << 26) // 2 - 1)
        t1 = util.make_text([g])
        d1, d1tw = h1.sort_text(t1)
        b1 = h1.dev_download(d1, t1.size)
        h1.dev_free(d1), h1.dev_free(d1tw)
        h1.from_plain(b1)
        st = rng.integers(0, len(g) - 150, size=reads_per_gpu)
        t2 = _reads_text(g, st, rng)
        d2, d2tw = h1.sort_text(t2)
        sent = np.flatnonzero(t2 == 0).astype(np.int64)
        from ropebwt3_amd import CommGroup
        bounds = interval_bounds(b1.size, 1)
        res = {}
        for driver in ("library", "python"):   # rb3gpu_sh_merge (one kernel + one read-back per round) against the loop in Python (rounds 1-3)
            if driver == "library":
                grp = CommGroup(1)
                comm, fn = grp.comm(0, h1), merge_interval_c
            else:
                comm, fn = SoloComm(dev, sync=lambda: (h1.sync(), torch.cuda.synchronize())), merge_interval
            stt = {}
            for _ in range(max(1, args.warmup)):
                fn(h1, comm, bounds, d2, d2tw, t2.size, sent, commit=False, stats=stt)
            h1.sync(), torch.cuda.synchronize()
            h1.stats_reset()
            t = time.perf_counter()
            for _ in range(args.steps):
                fn(h1, comm, bounds, d2, d2tw, t2.size, sent, commit=False, stats=stt)
            h1.sync(), torch.cuda.synchronize()
            dt = time.perf_counter() - t
            s1 = h1.stats()
            res[driver] = {"value": round(t2.size * args.steps / dt / 1e9, 6), "unit": "Gbp/s", "ms_per_step": round(dt / args.steps * 1e3, 4), "rounds": stt.get("rounds"),
                           "us_per_round_walk": round(s1["ms_rank"] / args.steps / max(1, stt.get("rounds") or 1) * 1e3, 2), "ms_rebuild": round(s1["ms_build"] / args.steps, 3)}
        # ... and through the NORMAL single-GPU path (rb3gpu_merge_text_dev, one walker per string): the number an interval-sharded build has to beat
        try:
            for _ in range(max(1, args.warmup)):
                h1.merge_text_dev(d2, d2tw, t2.size, int(sent.size), commit=False)
            h1.sync(), torch.cuda.synchronize()
            h1.stats_reset()
            t = time.perf_counter()
            for _ in range(args.steps):
                h1.merge_text_dev(d2, d2tw, t2.size, int(sent.size), commit=False)
            h1.sync(), torch.cuda.synchronize()
            dt = time.perf_counter() - t
            s1 = h1.stats()
            res["normal"] = {"value": round(t2.size * args.steps / dt / 1e9, 6), "unit": "Gbp/s", "ms_per_step": round(dt / args.steps * 1e3, 4), "ms_rank": round(s1["ms_rank"] / args.steps, 3), "ms_rebuild": round(s1["ms_build"] / args.steps, 3),
                             "entry_point": "rb3gpu_merge_text_dev, one walker per string (k_chain), commit=0"}
        except Exception as e:
            res["normal"] = {"error": repr(e)[:200]}
        h1.dev_free(d2), h1.dev_free(d2tw)
        out = dict(res["library"])
        out.update({"symbols_per_step": int(t2.size), "index_symbols": int(b1.size), "loop_in_python": res["python"], "normal_single_gpu_path": res["normal"],
                    "note": "rank 0 alone, same protocol with one interval (no collective), measured after the timed region of this run; us_per_round_walk: the walk's wall time per lock-step round (kernel + read-back of the split sizes), rebuild excluded"})
        return out
    finally:
        h1.close()


def interval_leg(args, h, dist, rank, world, dev, local_rank, shared_gpu, barrier):
    """the north_star workload: the index of a random genome cut into `world` intervals, one per GPU; every step merges one batch of reads
    (world x 500 k reads of 150 bp, both strands: per-GPU work fixed) with the BATCH sharded as well (rb3gpu_sh_merge_text: 1 byte per batch
    symbol replicated, the text-order words by text range).  Returns rank 0's JSON object (None elsewhere)."""
    import os, time
    import torch
    from tests import util
    out = None
    n_index, reads_per_gpu = (1 << 26) * world, (500000 if world > 1 else 100000)   # (large batches amortise the per-symbol collectives: config 4 has 46 M chains per batch)
    if shared_gpu and world > 2:   # TEST MODE: every rank sorts and holds the whole batch on the ONE device (~100 B per symbol and rank): four ranks ran out of memory at 604 M symbols
        reads_per_gpu = min(reads_per_gpu, 5000000 // (world * world))
        n_index = (1 << 24) * world   # (... and the index text as well)
    rng = np.random.default_rng(31)
    g = util.random_genome(rng, n_index // 2 - 1)
    t1 = util.make_text([g])
    d1, d1tw = h.sort_text(t1)                                  # every rank sorts the (synthetic) index text itself: no broadcast needed
    b1 = h.dev_download(d1, t1.size)
    h.dev_free(d1), h.dev_free(d1tw)
    bounds = interval_bounds(b1.size, world)
    h.from_plain(b1[bounds[rank]:bounds[rank + 1]])
    st = rng.integers(0, len(g) - 150, size=reads_per_gpu * world)
    t2 = _reads_text(g, st, rng)
    d2, d2tw = h.sort_text(t2)
    sent = np.flatnonzero(t2 == 0).astype(np.int64)
    driver = getattr(args, "sh_driver", None) or os.environ.get("RB3_SH_DRIVER", "rccl")
    if driver == "python":   # rounds 1-3: the lock-step loop in Python, two kernels and two host syncs per round
        comm, label, fn = TorchComm(dist, rank, world, dev, sync=lambda: (h.sync(), torch.cuda.synchronize())), "loop in Python (multi.merge_interval) over torch.distributed", merge_interval
    else:
        comm, label = sh_comm(h, dist, rank, world, dev, driver)
        fn = merge_interval_c
    a1, a2 = d2, d2tw
    if fn is merge_interval_c and not os.environ.get("RB3_SH_REPLICATED_BATCH"):   # the batch sharded as well: what `build --gpus N --interval` runs
        n2 = t2.size
        t_lo = n2 // world * rank + (n2 % world) * rank // world
        a1, a2 = h.tprev_from_tw(d2tw, n2), d2tw.value + t_lo * 8   # (every rank sorted the whole synthetic batch itself; it only reads its own text range of the words)
        fn = merge_interval_text
    stt = {}
    for _ in range(args.warmup):
        fn(h, comm, bounds, a1, a2, t2.size, sent, commit=False, stats=stt)
    h.stats_reset()
    barrier()
    t = time.perf_counter()
    for _ in range(args.steps):
        fn(h, comm, bounds, a1, a2, t2.size, sent, commit=False, stats=stt)
    barrier()
    dt = time.perf_counter() - t
    tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if shared_gpu else dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    s = h.stats()
    if rank == 0:
        out = {"metric": "Gbp/s indexed (build merge)", "value": round(t2.size * args.steps / dt / 1e9, 6), "unit": "Gbp/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64", "data": "synthetic",
               "config": {"workload": "interval-sharded index (north_star): %d symbols in %d intervals, one per GPU; per step one batch of %d x 150 bp reads (both strands, %d symbols) merged with one all-to-all per symbol" % (b1.size, world, reads_per_gpu * world, t2.size),
                          "symbols_per_step_per_gpu": int(t2.size // world), "index_symbols": int(b1.size), "parallelism": "interval%d: chain states routed to the owner of their insertion point by %s, %d lock-step rounds per merge; local rebuild per interval" % (world, "gloo send/recv through host memory (ranks share a GPU: TEST MODE, not a measurement)" if shared_gpu else "RCCL all-to-all(v)", stt.get("rounds", 0)),
                          "rows_per_rank": stt.get("rows_per_rank"), "ranks_share_a_gpu": bool(shared_gpu),
                          "driver": ("rb3gpu_sh_merge_text (the loop inside the library; the batch sharded: 1 byte per symbol on every GPU, the text-order words by text range, rows looked up by their owners at the end): " + label) if fn is merge_interval_text else ("rb3gpu_sh_merge (the loop inside the library): " + label) if fn is merge_interval_c else label,
                          "us_per_round_rank0": round(s["ms_rank"] / args.steps / max(1, stt.get("rounds") or 1) * 1e3, 2)},
               "phases_ms_per_step_rank0": {"step_kernels": round(s["ms_rank"] / args.steps, 3), "rebuild": round(s["ms_build"] / args.steps, 3)},
               "roofline": {"bound": "hbm", "kernel": "k_sh_round" if fn is not merge_interval else "k_sh_step", "achieved": round(208 * t2.size / world * args.steps / max(1e-9, s["ms_rank"]) / 1e6, 1), "peak": 8000.0, "unit": "GB/s",
                            "frac": round(208 * t2.size / world * args.steps / max(1e-9, s["ms_rank"]) / 1e6 / 8000.0, 5), "traffic": None,
                            "note": "rank 0: 208 B x the LF steps it executed / the time of its step kernels; the collectives are outside this figure and inside `value`"}}
        # the same per-GPU work on ONE GPU (rank 0 alone, its own handle, no collectives): what weak scaling is measured against
        if world > 1 or os.environ.get("RB3_BENCH_SOLO_REF"):
            try:
                out["n1_same_workload"] = _solo_interval_reference(reads_per_gpu, args, dev, local_rank)
            except Exception as e:   # never lose the measurement above to this extra
                out["n1_same_workload"] = {"error": repr(e)}

    return out

def _emit(obj):
    """the one JSON line of bench.py, on the real stdout (bench.protect_stdout)"""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    bench.emit_json(obj)


def bench_main(args, rank, local_rank, world):
    """bench.py --gpus N (N > 1), one process per GPU.  --mode interval (default): north_star -- the index of a random genome
    is cut into N intervals, every step merges one batch of reads (N x 100 k reads of 150 bp, both strands: per-GPU work fixed)
    with merge_interval; the timed step holds every all-gather, the all-to-all of every symbol and the local rebuilds.
    --mode partition (default): bench_partition_mtb below -- the headline workload (mtb152), partitioned + tree merge.
    --mode replicated: one batch of N genomes, walkers sharded by text range, all-reduce of pos[]."""
    import json
    import os
    import sys
    import time
    import torch
    import torch.distributed as dist
    from ropebwt3_amd import Rb3Gpu, host
    from tests import util
    if args.mode == "partition":
        return bench_partition_mtb(args, rank, local_rank, world)
    ndev = torch.cuda.device_count()
    dev_id = local_rank % ndev
    shared_gpu = ndev < int(os.environ.get("LOCAL_WORLD_SIZE", world))   # fewer devices than ranks: TEST MODE, the ranks share GPUs and talk over gloo
    torch.cuda.set_device(dev_id)
    dev = torch.device("cuda", dev_id)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if shared_gpu:
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        if args.mode == "interval":
            args.sh_driver = "gloo"
    else:
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
    local_rank = dev_id
    h = Rb3Gpu(device=local_rank, verbose=1)

    def barrier():
        h.sync()
        torch.cuda.synchronize()
        dist.barrier()

    out = None
    if args.mode == "interval":
        out = interval_leg(args, h, dist, rank, world, dev, local_rank, shared_gpu, barrier)
    else:
        seeds = [2 + i for i in range(world)]
        g0 = util.random_genome(np.random.default_rng(1), args.genome_len)
        gs = [util.mutate(np.random.default_rng(sd), g0, args.div) for sd in seeds]
        b1 = host.build_bwt(util.make_text([g0]))
        b2, walkers = host.build_bwt_walkers(util.make_text(gs), args.walker_step)
        h.from_plain(b1)
        d2 = h.dev_upload(b2)
        pos = torch.empty(b2.size, dtype=torch.int64, device=dev)
        for _ in range(args.warmup):
            merge_sharded(h, d2, b2.size, walkers, args.walker_step, dist, rank, world, pos, commit=False, sync=torch.cuda.synchronize)
        barrier()
        t = time.perf_counter()
        for _ in range(args.steps):
            merge_sharded(h, d2, b2.size, walkers, args.walker_step, dist, rank, world, pos, commit=False, sync=torch.cuda.synchronize)
        barrier()
        dt = time.perf_counter() - t
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        if rank == 0:
            out = {"metric": "Gbp/s indexed (build merge)", "value": round(b2.size * args.steps / dt / 1e9, 6), "unit": "Gbp/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                   "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64", "data": "synthetic",
                   "config": {"workload": "one batch of %d genomes of %d bp merged into the index of G0, replicated on every GPU" % (world, args.genome_len),
                              "parallelism": "replicated%d: walkers sharded by text range, all-reduce(MAX) of pos[] (8 B per batch row), every GPU rebuilds its replica" % world}}
    if rank == 0:
        _emit(out)
    h.close()
    dist.destroy_process_group()


def bench_partition_mtb(args, rank, local_rank, world):
    """bench.py --gpus N, default mode: the SAME job as the N = 1 headline -- the mtb152 build, md5-gated -- with the input
    partitioned: rank r builds the index of genomes [K r / N, K (r+1) / N) exactly as the single-GPU build does (one genome per
    batch; H2D + merges timed, suffix sorting not counted), then the N indexes are combined into rank 0's by a binary tree of
    whole-index merges (plain BWTs over RCCL/xGMI, merged through the reference's signature, rb3gpu_merge_plain_dev).
    One step = the whole partitioned build: max over ranks of the leaf merge paths + the tree merge (barriers on both sides).
    `value` = the symbols the single-GPU build merges (everything but the first genome) / that time: the job and its output
    (the .fmd, byte-identical to the reference's) are the same for every N, so the curve over N is strong scaling.

    Where several ranks share a GPU (a one-GPU box: RCCL refuses two ranks on one device) the tree's transfers go through
    host memory over gloo -- the same code path otherwise; that mode is for testing, not for numbers."""
    import hashlib
    import json
    import os
    import sys
    import time
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench as B
    ndev = torch.cuda.device_count()
    dev_id = local_rank % ndev
    shared_gpu = ndev < int(os.environ.get("LOCAL_WORLD_SIZE", world))
    torch.cuda.set_device(dev_id)
    dev = torch.device("cuda", dev_id)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if shared_gpu:
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
    K, L = args.mtb, args.genome_len
    gold = B.mtb_manifest(K, L)
    # rank 0 writes the files (a node-local tmpfs), everybody reads its slice
    box = [None]
    if rank == 0:
        tmp, files, _ = B.mtb_files(K, L)
        box[0] = (tmp, files)
    dist.broadcast_object_list(box, src=0)
    tmp, files = box[0]
    lo, hi = K * rank // world, K * (rank + 1) // world
    texts, walkers, keep = B.load_batches(files[lo:hi], pinned=not args.no_pinned, device=dev_id)
    bl = B.BuildLoop(dev_id)
    link = TreeLink(dist, dev, not shared_gpu, sync=torch.cuda.synchronize)
    zero = torch.zeros(1, device=dev) if not shared_gpu else torch.zeros(1)

    def barrier():
        bl.h.sync()
        torch.cuda.synchronize()
        dist.all_reduce(zero)     # (a collective on the data path's own transport; dist.barrier() on nccl needs a device guess)
        torch.cuda.synchronize()

    def step():
        a, b, c, n, w = bl.run(texts, walkers)
        barrier()
        t = time.perf_counter()
        tree_merge(bl.h, dist, rank, world, dev, sync=torch.cuda.synchronize, link=link)
        barrier()
        return a + b, time.perf_counter() - t, c, w

    for _ in range(args.warmup):
        step()
    barrier()
    bl.h.stats_reset()
    leaf = tree = sort = 0.0
    for _ in range(args.steps):
        a, t, c, w = step()
        leaf, tree, sort = leaf + a, tree + t, sort + c
    tt = torch.tensor([leaf, tree, sort], dtype=torch.float64, device=dev if not shared_gpu else "cpu")
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    leaf, tree, sort = (float(x) for x in tt)
    st = bl.h.stats()
    if rank == 0:
        md5, fmd_len = bl.fmd_md5()
        first = os.path.getsize(files[0])  # (not used for the numbers)
        tot = bl.h.get_tot()
        n0 = int(texts[0].size)
        nsym = tot - n0
        dt = (leaf + tree) / args.steps
        ident = (md5 == gold["fmd_md5"]) if gold else None
        out = {"metric": "Gbp/s indexed (build merge)", "value": round(nsym / dt / 1e9, 6), "unit": "Gbp/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "int64", "data": "synthetic",
               "config": {"workload": "cfg3-synthetic-mtb%d: the same %d genomes of %d bp as the single-GPU headline, %d symbols merged per step; input partitioned over %d GPUs (%d-%d genomes each, one genome per batch), per-GPU indexes combined by a binary tree of whole-index merges" % (K, K, L, nsym, world, K // world, -(-K // world)),
                          "symbols_per_step": int(nsym), "index_symbols_final": int(tot), "parallelism": "partition%d + tree merge: leaves = the single-GPU merge path on a slice (H2D + rb3gpu_merge_text_dev per genome); tree = rb3gpu_export_plain_dev -> %s -> rb3gpu_merge_plain_dev, %d levels" % (world, "send/recv over gloo through host memory (ranks share a GPU: test mode)" if shared_gpu else "RCCL send/recv over xGMI", (world - 1).bit_length()),
                          "fmd_md5": md5, "fmd_bytes": fmd_len, "fmd_identical_to_reference": ident, "ranks_share_a_gpu": bool(shared_gpu)},
               "phases_ms_per_step": {"leaf_merge_paths(max over ranks)": round(leaf / args.steps * 1e3, 3), "tree_merge": round(tree / args.steps * 1e3, 3)},
               "not_counted_ms_per_step": {"suffix_sorting_on_the_gpu(max over ranks)": round(sort / args.steps * 1e3, 3)},
               "roofline": B.chain_roofline(int(st["n_lf_steps"] / max(1, st["n_rank_launches"])), st["ms_chain"] / max(1, st["n_rank_launches"]), "text", None,
                                            "rank 0's k_chain launches (leaf rounds and tree merges together): 208 B x LF steps / HIP-event time")}
    # in the same line, the reads workload through the interval-sharded path (north_star), with the SAME workload at N = 1 through
    # the normal single-GPU path beside it (aux_interval_reads.n1_same_workload.normal_single_gpu_path: the number --interval has to beat)
    aux = None
    # The leg below has never run between two devices (no multi-GPU box in rounds 1-5): if it hangs in a collective, the line of the main
    # measurement must not be lost with it.  Every rank arms the same deadline; when it passes, rank 0 prints the line without the leg and all exit.
    watchdog = None
    if not args.no_aux:
        import threading

        printed = [False]
        emit_lock = threading.Lock()   # the line is written exactly once: by the deadline or by the main thread, never by both and never by neither

        def give_up():
            with emit_lock:
                if rank == 0 and not printed[0]:
                    out["aux_interval_reads"] = {"error": "TIMED OUT: the interval leg did not finish within %d s (first contact with a multi-GPU box?); the line above it stands" % aux_deadline, "timed_out": True}
                    _emit(out)
                    printed[0] = True
                os._exit(3 if rank != 0 else 0)   # (rank 0 has delivered the main measurement; the other ranks report that they were cut off)

        aux_deadline = int(os.environ.get("RB3_BENCH_AUX_DEADLINE", "420"))
        watchdog = threading.Timer(aux_deadline, give_up)
        watchdog.daemon = True
        watchdog.start()
    if not args.no_aux:
        try:
            from ropebwt3_amd import Rb3Gpu
            h2 = Rb3Gpu(device=dev_id, verbose=1)
            if shared_gpu:
                args.sh_driver = "gloo"
            aux = interval_leg(args, h2, dist, rank, world, dev, dev_id, shared_gpu, lambda: (h2.sync(), torch.cuda.synchronize(), dist.all_reduce(zero), torch.cuda.synchronize()))
            h2.close()
        except Exception as e:
            aux = {"error": repr(e)[:300]}
    if rank == 0:
        if aux is not None:
            out["aux_interval_reads"] = aux
        if watchdog is not None:
            with emit_lock:
                if not printed[0]:
                    _emit(out)
                    printed[0] = True
        else:
            _emit(out)
    dist.barrier() if shared_gpu else barrier()   # (a rank whose leg failed waits here for the others -- or for the deadline: the timer stays armed, it no longer writes anything)
    if watchdog is not None:
        watchdog.cancel()
    bl.close()
    if rank == 0:
        for f in files:
            try:
                os.unlink(f)
            except OSError:
                pass
        try:
            os.rmdir(tmp)
        except OSError:
            pass
    dist.destroy_process_group()
