"""Test helpers: ctypes bindings of the CPU oracle (oracle/liboracle.so), of the unmodified
reference build (oracle/_ref/librb3ref.so, when present) and seeded input generators.

Test infrastructure only -- nothing under ropebwt3_amd/ imports this.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "librb3ref.so")
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "ropebwt3")
GOLDEN = os.path.join(ROOT, "tests", "golden")

SYMS = "$ACGTN"


def sym_str(b):
    return "".join(SYMS[x] for x in b)


class Oracle:
    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
        L = ctypes.CDLL(ORACLE_SO)
        L.orc_bwt.restype = ctypes.c_int64
        L.orc_bwt.argtypes = [ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
        L.orc_text_from_lines.restype = ctypes.c_int64
        L.orc_text_from_lines.argtypes = [ctypes.c_int64, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.orc_merge_plain.restype = ctypes.c_int
        L.orc_merge_plain.argtypes = [ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.orc_mg_rank_plain.restype = ctypes.c_int
        L.orc_mg_rank_plain.argtypes = [ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.orc_runs.restype = ctypes.c_int64
        L.orc_runs.argtypes = [ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
        L.orc_ssa_dims.restype = None
        L.orc_ssa_dims.argtypes = [ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int)]
        L.orc_ssa_gen.restype = ctypes.c_int
        L.orc_ssa_gen.argtypes = [ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        self.L = L

    def ssa_gen(self, b, ss):
        """(ms, r2i, ssa) of the index whose plain BWT is b (ssa.c:54-81)"""
        b = np.ascontiguousarray(b, dtype=np.uint8)
        m, n_ssa, ms = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int()
        self.L.orc_ssa_dims(b.size, b.ctypes.data, ss, ctypes.byref(m), ctypes.byref(n_ssa), ctypes.byref(ms))
        r2i = np.zeros(max(m.value, 1), dtype=np.uint64)
        ssa = np.zeros(max(n_ssa.value, 1), dtype=np.uint64)
        self.L.orc_ssa_gen(b.size, b.ctypes.data, ss, r2i.ctypes.data, ssa.ctypes.data)
        return ms.value, r2i[:m.value], ssa[:n_ssa.value]

    def text(self, lines, fwd=True, rev=True):
        s = ("\n".join(lines) + "\n").encode()
        out = np.zeros(2 * (len(s) + 1), dtype=np.uint8)
        l = self.L.orc_text_from_lines(len(s), s, int(fwd), int(rev), out.ctypes.data)
        return out[:l].copy()

    def bwt(self, text):
        text = np.ascontiguousarray(text, dtype=np.uint8)
        b = np.zeros(text.size, dtype=np.uint8)
        n = self.L.orc_bwt(text.size, text.ctypes.data, b.ctypes.data)
        assert n > 0, "orc_bwt rejected the text"
        return b

    def merge(self, b1, b2, threads=4):
        b1 = np.ascontiguousarray(b1, dtype=np.uint8)
        b2 = np.ascontiguousarray(b2, dtype=np.uint8)
        out = np.zeros(b1.size + b2.size, dtype=np.uint8)
        r = self.L.orc_merge_plain(b1.size, b1.ctypes.data, b2.size, b2.ctypes.data, out.ctypes.data, threads)
        assert r == 0, "orc_merge_plain returned %d" % r
        return out

    def mg_rank(self, b1, b2, threads=4):
        b1 = np.ascontiguousarray(b1, dtype=np.uint8)
        b2 = np.ascontiguousarray(b2, dtype=np.uint8)
        rb = np.zeros(b2.size, dtype=np.int64)
        acc2 = np.zeros(7, dtype=np.int64)
        r = self.L.orc_mg_rank_plain(b1.size, b1.ctypes.data, b2.size, b2.ctypes.data, rb.ctypes.data, acc2.ctypes.data, threads)
        assert r == 0
        return rb, acc2

    def runs(self, b):
        b = np.ascontiguousarray(b, dtype=np.uint8)
        n = self.L.orc_runs(b.size, b.ctypes.data, None)
        r = np.zeros(n, dtype=np.int64)
        self.L.orc_runs(b.size, b.ctypes.data, r.ctypes.data)
        return [(int(x & 7), int(x >> 3)) for x in r]


class Reference:
    """The unmodified reference, compiled by oracle/Makefile into oracle/_ref/ (may be absent)."""

    def __init__(self):
        if not os.path.exists(REF_SO):
            raise FileNotFoundError(REF_SO)
        L = ctypes.CDLL(REF_SO)
        L.rb3_build_sais.restype = None
        L.rb3_build_sais.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int]
        L.rb3_enc_plain2fmr.restype = ctypes.c_void_p
        L.rb3_enc_plain2fmr.argtypes = [ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int32]
        L.rb3_fmi_merge_plain.restype = None
        L.rb3_fmi_merge_plain.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int]
        L.mr_destroy.restype = None
        L.mr_destroy.argtypes = [ctypes.c_void_p]
        try:
            ctypes.c_int.in_dll(L, "rb3_verbose").value = 0
        except ValueError:
            pass
        self.L = L

    @staticmethod
    def available():
        return os.path.exists(REF_SO)

    def mg_rank(self, b1, b2, threads=2):
        """the reference's own rb[] (rb3_mg_rank_plain, fm-index.c:202-225, on the mrope of b1 built by rb3_enc_plain2fmr):
        rb[kb] = (ka + kb) << 6 | B2[kb] << 3 | first symbol of the row's suffix, and acc2 = C array of b2"""
        class Fmi(ctypes.Structure):   # rb3_fmi_t, fm-index.h:42-49
            _fields_ = [("is_fmd", ctypes.c_int32), ("e", ctypes.c_void_p), ("r", ctypes.c_void_p), ("ssa", ctypes.c_void_p), ("sid", ctypes.c_void_p), ("acc", ctypes.c_int64 * 7)]
        L = self.L
        L.rb3_fmi_get_acc.restype = ctypes.c_int64
        L.rb3_fmi_get_acc.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.rb3_mg_rank_plain.restype = None
        L.rb3_mg_rank_plain.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        b1 = np.ascontiguousarray(b1, dtype=np.uint8)
        b2 = np.ascontiguousarray(b2, dtype=np.uint8)
        r = L.rb3_enc_plain2fmr(b1.size, b1.ctypes.data, 0, 0, threads)
        f = Fmi()
        f.is_fmd, f.e, f.r, f.ssa, f.sid = 0, None, r, None, None
        L.rb3_fmi_get_acc(ctypes.byref(f), f.acc)     # rb3_fmi_init, fm-index.h:95-101
        rb = np.zeros(b2.size, dtype=np.int64)
        acc2 = np.zeros(7, dtype=np.int64)
        L.rb3_mg_rank_plain(ctypes.byref(f), b2.size, b2.ctypes.data, rb.ctypes.data, acc2.ctypes.data, threads)
        L.mr_destroy(r)
        return rb, acc2

    def bwt(self, text, threads=4):
        t = np.ascontiguousarray(text, dtype=np.uint8).copy()
        n_seq = int((t == 0).sum())
        self.L.rb3_build_sais(n_seq, t.size, t.ctypes.data, threads)
        return t


def ssa_bytes(ss, ms, r2i, ssa):
    """the bytes rb3_ssa_dump writes (ssa.c:198-213)"""
    import struct
    return b"SSA\1" + struct.pack("<IIqq", ss, ms, len(r2i), len(ssa)) + np.asarray(r2i, dtype="<u8").tobytes() + np.asarray(ssa, dtype="<u8").tobytes()


def random_genome(rng, n):
    return rng.integers(1, 5, size=n, dtype=np.uint8)


def mutate(rng, g, rate):
    g = g.copy()
    k = int(len(g) * rate)
    if k:
        idx = rng.choice(len(g), size=k, replace=False)
        g[idx] = ((g[idx] - 1 + rng.integers(1, 4, size=k)) % 4 + 1).astype(np.uint8)
    return g


def revcomp(s):
    r = s[::-1].copy()
    m = (r >= 1) & (r <= 4)
    r[m] = 5 - r[m]
    return r


def make_text(seqs, fwd=True, rev=True):
    """io.c:84-102: every record contributes its forward strand then its reverse complement,
    each followed by a 0."""
    parts = []
    z = np.zeros(1, dtype=np.uint8)
    for s in seqs:
        s = np.asarray(s, dtype=np.uint8)
        if fwd:
            parts += [s, z]
        if rev:
            parts += [revcomp(s), z]
    return np.concatenate(parts)


def reads_from(rng, genome, n_reads, read_len, err=0.0):
    st = rng.integers(0, len(genome) - read_len + 1, size=n_reads)
    out = []
    for s in st:
        r = genome[s:s + read_len].copy()
        if err > 0:
            m = rng.random(read_len) < err
            r[m] = rng.integers(1, 5, size=int(m.sum()), dtype=np.uint8)
        out.append(r)
    return out
