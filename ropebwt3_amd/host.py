"""ctypes binding of the host-side C library (librb3host.so): per-batch suffix sorting and the
FMD/FMR codecs.  These are the host halves of `ropebwt3 build` (sais-ss.c, rld0.c, mrope.c/rope.c
dump format) written from scratch in C; Python only marshals buffers."""
import ctypes
import os

import numpy as np

from . import _build

_lib = None
RUN_F = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64)


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_build.LIB_HOST):
        raise RuntimeError("%s is missing: run __graft_entry__.build()" % _build.LIB_HOST)
    L = ctypes.CDLL(os.environ.get("RB3HOST_LIB", _build.LIB_HOST))  # (override for experiments only)
    L.rb3h_build_bwt.restype = ctypes.c_int
    L.rb3h_build_bwt.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int]
    L.rb3h_build_bwt_walkers.restype = ctypes.c_int
    L.rb3h_build_bwt_walkers.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                         ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_void_p)]
    L.rb3h_fmdw_init.restype = ctypes.c_void_p
    L.rb3h_fmdw_enc.restype = ctypes.c_int
    L.rb3h_fmdw_enc.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int]
    L.rb3h_fmdw_finish.restype = ctypes.c_int
    L.rb3h_fmdw_finish.argtypes = [ctypes.c_void_p]
    L.rb3h_fmdw_destroy.restype = None
    L.rb3h_fmdw_destroy.argtypes = [ctypes.c_void_p]
    L.rb3h_seq_open.restype = ctypes.c_void_p
    L.rb3h_seq_open.argtypes = [ctypes.c_char_p, ctypes.c_int]
    L.rb3h_seq_close.restype = None
    L.rb3h_seq_close.argtypes = [ctypes.c_void_p]
    L.rb3h_seq_read.restype = ctypes.c_int64
    L.rb3h_seq_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int64)]
    L.rb3h_parse_num.restype = ctypes.c_int64
    L.rb3h_parse_num.argtypes = [ctypes.c_char_p]
    L.free.argtypes = [ctypes.c_void_p] if hasattr(L, "free") else None
    _lib = L
    return L


class _Buf(ctypes.Structure):
    _fields_ = [("l", ctypes.c_int64), ("m", ctypes.c_int64), ("s", ctypes.c_void_p)]


def build_bwt(text, n_threads=1):
    """text: uint8 array of 0-terminated nt6 strings -> its BWT (rb3_build_sais, sais-ss.c:50-56)"""
    L = load_library()
    t = np.ascontiguousarray(text, dtype=np.uint8).copy()
    r = L.rb3h_build_bwt(0, t.size, t.ctypes.data, n_threads)
    if r < 0:
        raise ValueError("rb3h_build_bwt failed with code %d" % r)
    return t


def build_bwt_walkers(text, step=1024, n_threads=1):
    """BWT of the batch plus its LF-walker list for Rb3Gpu.merge_plain_walkers: an (n, 4) int64
    array of (row, ka0, nsteps, flags), one walker per string and one per `step` text positions."""
    L = load_library()
    t = np.ascontiguousarray(text, dtype=np.uint8).copy()
    nw, pw = ctypes.c_int64(0), ctypes.c_void_p()
    r = L.rb3h_build_bwt_walkers(0, t.size, t.ctypes.data, n_threads, step, ctypes.byref(nw), ctypes.byref(pw))
    if r < 0:
        raise ValueError("rb3h_build_bwt_walkers failed with code %d" % r)
    w = np.ctypeslib.as_array(ctypes.cast(pw, ctypes.POINTER(ctypes.c_int64)), shape=(nw.value, 4)).copy()
    libc = ctypes.CDLL(None)
    libc.free.argtypes = [ctypes.c_void_p]
    libc.free(pw)
    return t, w


def walkers_from_ckrow(text, step, ckrow):
    """the same walker list from a sampled inverse suffix array (ckrow[i] = row of text position i * step),
    e.g. the one Rb3Gpu.bwt_from_text returns; `text` is the batch text (not its BWT)"""
    L = load_library()
    L.rb3h_walkers_from_ckrow.restype = ctypes.c_int
    L.rb3h_walkers_from_ckrow.argtypes = [ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_void_p)]
    t = np.ascontiguousarray(text, dtype=np.uint8)
    ck = np.ascontiguousarray(ckrow, dtype=np.int64)
    nw, pw = ctypes.c_int64(0), ctypes.c_void_p()
    r = L.rb3h_walkers_from_ckrow(t.size, t.ctypes.data, step, ck.ctypes.data, ctypes.byref(nw), ctypes.byref(pw))
    if r < 0:
        raise ValueError("rb3h_walkers_from_ckrow failed with code %d" % r)
    w = np.ctypeslib.as_array(ctypes.cast(pw, ctypes.POINTER(ctypes.c_int64)), shape=(nw.value, 4)).copy()
    libc = ctypes.CDLL(None)
    libc.free.argtypes = [ctypes.c_void_p]
    libc.free(pw)
    return w


def walkers_text(text, step):
    """the walker list of a batch by TEXT POSITION (Rb3Gpu.merge_text_dev): one walker per string, at its sentinel,
    and one per `step` text positions; `text` is the batch text (not its BWT)"""
    L = load_library()
    L.rb3h_walkers_text.restype = ctypes.c_int
    L.rb3h_walkers_text.argtypes = [ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_void_p)]
    t = np.ascontiguousarray(text, dtype=np.uint8)
    nw, pw = ctypes.c_int64(0), ctypes.c_void_p()
    r = L.rb3h_walkers_text(t.size, t.ctypes.data, step, ctypes.byref(nw), ctypes.byref(pw))
    if r < 0:
        raise ValueError("rb3h_walkers_text failed with code %d" % r)
    w = np.ctypeslib.as_array(ctypes.cast(pw, ctypes.POINTER(ctypes.c_int64)), shape=(nw.value, 4)).copy()
    libc = ctypes.CDLL(None)
    libc.free.argtypes = [ctypes.c_void_p]
    libc.free(pw)
    return w


def read_batches(path, is_line, max_len, fwd=True, rev=True, byte_range=None):
    """Iterate the batches `build` would cut from one file (rb3_seq_read, io.c:104-125):
    yields (n_strings, text) with text a uint8 array.  byte_range = (beg, end): only the records that start in that byte
    range of a plain file (rb3h_seq_open_range, what a slice of `build --gpus N` reads; end 0 = to the end of the file)."""
    L = load_library()
    if byte_range is None:
        fp = L.rb3h_seq_open(path.encode(), int(is_line))
    else:
        L.rb3h_seq_open_range.restype = ctypes.c_void_p
        L.rb3h_seq_open_range.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64]
        fp = L.rb3h_seq_open_range(path.encode(), int(is_line), int(byte_range[0]), int(byte_range[1]))
    if not fp:
        raise IOError("cannot open %s" % path)
    libc = ctypes.CDLL(None)
    libc.free.argtypes = [ctypes.c_void_p]
    try:
        while True:
            b = _Buf(0, 0, None)
            ne = ctypes.c_int64(0)
            n = L.rb3h_seq_read(fp, ctypes.byref(b), max_len, int(fwd), int(rev), ctypes.byref(ne))
            if n < 0:   # no memory, or the file could not be read to its end (I/O error, truncated gzip stream)
                if b.s:
                    libc.free(b.s)
                raise ValueError("rb3h_seq_read failed with code %d" % n)
            if n <= 0 and b.l == 0:
                if b.s:
                    libc.free(b.s)
                if n < 0:
                    raise ValueError("parse error %d" % n)
                break
            arr = np.ctypeslib.as_array(ctypes.cast(b.s, ctypes.POINTER(ctypes.c_uint8)), shape=(b.l,)).copy()
            libc.free(b.s)
            yield int(n), arr
    finally:
        L.rb3h_seq_close(fp)


def strand_pairs(text, n_seq, max_pairs=32):
    """record offsets of a batch read with both strands (for Sorter.upload_fwd), or None if it has too many records / another layout"""
    L = load_library()
    L.rb3h_strand_pairs.restype = ctypes.c_int64
    L.rb3h_strand_pairs.argtypes = [ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
    ps = np.zeros(max_pairs, dtype=np.int64)
    n = L.rb3h_strand_pairs(text.size, text.ctypes.data, int(n_seq), max_pairs, ps.ctypes.data)
    return ps[:n].copy() if n > 0 else None


def parse_num(s):
    return int(load_library().rb3h_parse_num(s.encode()))


def fmd_bytes_from_plain(plain):
    """the .fmd of a BWT given as plain text (one byte per symbol: $ACGTN or nt6 codes; newline = sentinel), through the host FMD
    writer alone (rb3h_fmdw_enc / _finish: rld_enc / rld_enc_finish, rld0.c:137-216) -- the CPU-side test bench of the writer"""
    import tempfile
    L = load_library()
    L.rb3h_fmdw_dump_file.restype = ctypes.c_int
    L.rb3h_fmdw_dump_file.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
    lut = np.full(256, 5, dtype=np.uint8)
    for i, c in enumerate(b"$ACGTN"):
        lut[c] = i
    for i, c in enumerate(b"$acgtn"):
        lut[c] = i
    lut[10] = 0
    lut[:6] = np.arange(6)
    sym = lut[np.frombuffer(bytes(plain), dtype=np.uint8)]
    w = L.rb3h_fmdw_init()
    try:
        # runs, as rld_enc coalesces them anyway (rld0.c:153-161)
        cut = np.flatnonzero(np.diff(sym)) + 1
        starts = np.concatenate([[0], cut])
        lens = np.diff(np.concatenate([starts, [sym.size]]))
        for st, ln in zip(starts.tolist(), lens.tolist()):
            if L.rb3h_fmdw_enc(w, int(ln), int(sym[st])) < 0:
                raise ValueError("rb3h_fmdw_enc failed")
        L.rb3h_fmdw_finish(w)
        with tempfile.NamedTemporaryFile(suffix=".fmd") as f:
            if L.rb3h_fmdw_dump_file(w, f.name.encode()) < 0:
                raise IOError("rb3h_fmdw_dump_file failed")
            return open(f.name, "rb").read()
    finally:
        L.rb3h_fmdw_destroy(w)


def fmd_bytes_from_words(words, acc):
    """the whole .fmd file -- header, data section, rank index -- from the data section packed on the GPU
    (Rb3Gpu.export_fmd_words) and the index's C array: what `build -d` writes (rld_dump, rld0.c:222-243; the rank index is
    rld_rank_index, rld0.c:163-204, built by rb3h_fmdw_adopt on the host)"""
    import tempfile
    L = load_library()
    L.rb3h_fmdw_adopt.restype = ctypes.c_int
    L.rb3h_fmdw_adopt.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
    L.rb3h_fmdw_dump_file.restype = ctypes.c_int
    L.rb3h_fmdw_dump_file.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
    libc = ctypes.CDLL(None)
    libc.malloc.restype = ctypes.c_void_p
    libc.malloc.argtypes = [ctypes.c_size_t]
    libc.free.argtypes = [ctypes.c_void_p]
    words = np.ascontiguousarray(words, dtype=np.uint64)
    acc = np.ascontiguousarray(acc, dtype=np.int64)
    z = libc.malloc(words.nbytes)
    ctypes.memmove(z, words.ctypes.data, words.nbytes)
    w = L.rb3h_fmdw_init()
    if L.rb3h_fmdw_adopt(w, z, words.size, acc.ctypes.data) < 0:   # (on failure the array stays ours)
        libc.free(z)
        L.rb3h_fmdw_destroy(w)
        raise ValueError("rb3h_fmdw_adopt failed")
    with tempfile.NamedTemporaryFile(suffix=".fmd", dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as f:
        r = L.rb3h_fmdw_dump_file(w, f.name.encode())
        L.rb3h_fmdw_destroy(w)
        if r < 0:
            raise IOError("rb3h_fmdw_dump_file failed")
        return open(f.name, "rb").read()
