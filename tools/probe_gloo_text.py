"""the BATCH-sharded merge (rb3gpu_sh_merge_text) between PROCESSES over the gloo callbacks at several worlds, ranks sharing the GPU: does every interval come out as the oracle's?
(bench.py --gpus 8 in test mode failed in its interval leg with 'rows unset or misrouted' while the same merge between threads -- the CLI's --gpus 8 --interval -- was right)"""
import sys, os, socket
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def worker(rank, world, port, q, ipc, n_reads):
    import torch
    import torch.distributed as dist
    from ropebwt3_amd import Rb3Gpu, CallbackComm, multi, host, ipc_peer_enable
    from tests import util
    from tests import test_gpu_engine as T
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        orc = util.Oracle()
        rng = np.random.default_rng(99)
        g0 = util.random_genome(rng, 300000)
        cur = host.build_bwt(util.make_text([g0]))
        t2 = util.make_text(util.reads_from(rng, g0, n_reads, 75, err=0.01))
        want = orc.merge(cur, host.build_bwt(t2.copy()))
        bounds = multi.interval_bounds(cur.size, world)
        h = Rb3Gpu(verbose=1)

        def all_gather(vec):
            out = [torch.zeros(len(vec), dtype=torch.int64) for _ in range(world)]
            dist.all_gather(out, torch.from_numpy(np.ascontiguousarray(vec, dtype=np.int64)))
            return torch.stack(out).numpy()

        def exchange(d_send, stride, send_cnt, d_recv, recv_cnt):
            parts = []
            for d in range(world):
                n = int(send_cnt[d])
                parts.append(torch.from_numpy(h.dev_download_i64(d_send + d * stride * 16, n * 2)) if n else torch.zeros(0, dtype=torch.int64))
            recv = [torch.zeros(int(recv_cnt[s]) * 2, dtype=torch.int64) for s in range(world)]
            T._gloo_all_to_all(dist, rank, world, parts, recv)
            got = torch.cat(recv).numpy()
            if got.size:
                h.dev_upload_to(d_recv, got)

        comm = CallbackComm(rank, world, all_gather, exchange)
        if ipc:
            assert ipc_peer_enable(h, comm)
        h.from_plain(cur[bounds[rank]:bounds[rank + 1]])
        d_bwt, d_tw = h.sort_text(t2)
        n2 = t2.size
        t_lo = n2 // world * rank + (n2 % world) * rank // world
        d_tprev = h.tprev_from_tw(d_tw, n2)
        bounds, _ = h.sh_merge_text(comm, bounds, d_tprev, d_tw.value + t_lo * 8, n2, np.flatnonzero(t2 == 0))
        T._check_interval(h, np.random.default_rng(rank), want, bounds, rank)
        h.close()
        q.put((rank, True, ""))
    except BaseException as e:
        q.put((rank, False, repr(e)))
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    for world, ipc, n_reads in ((4, False, 20000), (8, False, 20000), (8, True, 20000), (8, False, 200000)):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=worker, args=(r, world, port, q, ipc, n_reads)) for r in range(world)]
        for p in procs:
            p.start()
        try:
            res = sorted(q.get(timeout=150) for _ in range(world))
        except Exception as e:
            res = [("timeout", False, repr(e))]
        for p in procs:
            p.join(timeout=5)
            if p.is_alive():
                p.terminate()
        print("world", world, "ipc", ipc, "reads", n_reads, "ok" if all(r[1] for r in res) else [(r[0], str(r[2])[:100]) for r in res if not r[1]], flush=True)
