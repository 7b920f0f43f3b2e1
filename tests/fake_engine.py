"""CPU stand-in for the four staged-merge calls of the HIP engine (mg_begin / mg_walk / pos /
mg_finish), for the gloo tests of ropebwt3_amd.multi.  It re-states the walker semantics of
k_chain<LIST> (rb3gpu_kernels.h) in numpy: bounds (lo, hi), exactness on lo == hi, stopping at the
end of the own segment when inexact, check mode, stop_row hand-off.  TEST INFRASTRUCTURE ONLY --
the product never imports it."""
import numpy as np

UNSET = -1


class FakeEngine:
    def __init__(self, b1):
        self.b1 = np.asarray(b1, dtype=np.uint8)
        n = self.b1.size
        self.occ = np.zeros((6, n + 1), dtype=np.int64)
        for c in range(6):
            self.occ[c, 1:] = np.cumsum(self.b1 == c)
        self.acc = np.concatenate([[0], np.cumsum(self.occ[:, n])])
        self.steps = 0

    def LF(self, c, k):
        return int(self.acc[c] + self.occ[c, k])

    def mg_begin(self, b2, length, pos_ptr_unused, pos_tensor=None):
        b2 = np.asarray(b2, dtype=np.uint8)
        assert b2.size == length
        self.b2 = b2
        cnt = np.bincount(b2, minlength=6)
        acc2 = np.concatenate([[0], np.cumsum(cnt)])
        self.acc2 = acc2
        occ = np.zeros(length, dtype=np.int64)
        seen = np.zeros(6, dtype=np.int64)
        for i, c in enumerate(b2):
            occ[i] = seen[c]
            seen[c] += 1
        self.lf2 = acc2[b2] + occ
        self.pos = pos_tensor.numpy() if pos_tensor is not None else np.full(length, UNSET, dtype=np.int64)
        self.pos[:] = UNSET
        return acc2

    def mg_walk(self, walkers, stop_row=-1):
        arrive = -1
        n1, m1 = self.b1.size, int(self.acc[1])
        for row, ka0, nsteps, flags in np.asarray(walkers, dtype=np.int64):
            kb, remaining, foreign = int(row), int(nsteps), bool(flags & 2)
            if ka0 == -2:
                ka0 = m1
            if ka0 >= 0:
                lo = hi = int(ka0)
            else:
                lo, hi = 0, n1
                if self.pos[kb] != UNSET:
                    continue
            while True:
                exact = lo == hi
                c = int(self.b2[kb])
                self.steps += 1
                if exact:
                    if foreign and self.pos[kb] != UNSET:
                        break
                    self.pos[kb] = lo + kb
                if c == 0:
                    break
                lo, hi = self.LF(c, lo), self.LF(c, hi)
                kb = int(self.lf2[kb])
                if kb == stop_row:
                    if lo == hi:
                        arrive = lo
                    break
                remaining -= 1
                if remaining == 0:
                    if lo != hi:
                        break
                    foreign, remaining = True, 1 << 62
        return arrive

    def mg_finish(self, commit):
        assert (self.pos != UNSET).all(), "rows left unset"
        assert (np.diff(self.pos) > 0).all(), "pos not increasing"


class FakePlainEngine:
    """index = its plain BWT; merge via the CPU oracle.  For the gloo test of multi.tree_merge."""

    def __init__(self, oracle, bwt):
        self.orc, self.b = oracle, np.asarray(bwt, dtype=np.uint8).copy()

    def get_tot(self):
        return int(self.b.size)

    def export_plain_dev(self, ptr):
        import ctypes
        ctypes.memmove(ptr, self.b.ctypes.data, self.b.size)

    def merge_plain_dev(self, ptr, n, commit=True):
        import ctypes
        b2 = np.empty(n, dtype=np.uint8)
        ctypes.memmove(b2.ctypes.data, ptr, n)
        self.b = self.orc.merge(self.b, b2)

    # the host-memory transport of multi.TreeLink (ranks that share one GPU)
    def export_plain(self):
        return self.b.copy()

    def merge_plain_host(self, b2):
        self.b = self.orc.merge(self.b, np.asarray(b2, dtype=np.uint8))


class FakeShardEngine:
    """numpy stand-in for the interval-sharded calls (rb3gpu_sh_step / rb3gpu_sh_finish / rb3gpu_get_acc) of ONE rank: it
    holds the symbols of its interval of the accumulated BWT.  For the gloo tests of ropebwt3_amd.multi.merge_interval."""

    def __init__(self, b1_slice):
        self.b = np.asarray(b1_slice, dtype=np.uint8).copy()
        self.steps = 0

    def get_acc(self):
        return np.concatenate([[0], np.cumsum(np.bincount(self.b, minlength=6)[:6])]).astype(np.int64)

    def sh_step(self, n_states, d_in, d_tw, d_ka, adj, bounds, my_iv, d_send):
        n_iv = len(bounds) - 1
        counts = np.zeros(n_iv + 1, dtype=np.int64)
        acc = self.get_acc()
        start = int(bounds[my_iv])
        out = [[] for _ in range(n_iv)]
        for q in range(n_states):
            tp, ka = int(d_in[q, 0]), int(d_in[q, 1])
            x = int(d_tw[tp])
            kb, c = x >> 3, x & 7
            k = ka - start
            assert 0 <= k <= self.b.size, "state routed to the wrong interval"
            d_ka[kb] = ka
            self.steps += 1
            if c == 0:
                counts[n_iv] += 1
                continue
            nka = int(acc[c] + np.count_nonzero(self.b[:k] == c) + adj[c])
            d = int(np.sum(np.asarray(bounds[1:n_iv]) <= nka))
            out[d].append((tp - 1, nka))
            counts[d] += 1
        at = 0
        for d in range(n_iv):
            for tp, ka in out[d]:
                d_send[at, 0], d_send[at, 1] = tp, ka
                at += 1
        return counts

    def sh_finish(self, jlo, n_rows, d_bwt, d_ka, iv_start, commit=True):
        if n_rows == 0:
            return
        ka = np.asarray(d_ka[jlo:jlo + n_rows], dtype=np.int64)
        assert (ka >= iv_start).all(), "rows of the interval unset or misrouted"
        assert (np.diff(ka) >= 0).all(), "ka not monotone"
        if commit:
            self.b = np.insert(self.b, ka - iv_start, np.asarray(d_bwt[jlo:jlo + n_rows], dtype=np.uint8))
