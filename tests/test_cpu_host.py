"""Host-side C (librb3host.so + the ropebwt3-amd CLI's host-only commands) against golden files
written by the reference and against the oracle.  No GPU needed."""
import gzip
import hashlib
import json
import os
import re
import subprocess

import numpy as np
import pytest

from ropebwt3_amd import _build, gpu, host
from tests import util

MAN = json.load(open(os.path.join(util.GOLDEN, "MANIFEST.json")))
CLI = _build.BIN_CLI


def run(cmd, inp=None):
    r = subprocess.run(cmd, input=inp, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-400:]
    return r.stdout


def test_sais_matches_oracle(oracle):
    rng = np.random.default_rng(7)
    for it in range(200):
        seqs = [rng.integers(1, 1 + int(rng.integers(1, 6)), size=int(rng.integers(1, 30)), dtype=np.uint8) for _ in range(int(rng.integers(1, 6)))]
        if it % 4 == 0:
            seqs += [seqs[0].copy(), util.revcomp(seqs[0])]
        t = util.make_text(seqs, rev=bool(it & 1))
        assert np.array_equal(host.build_bwt(t), oracle.bwt(t))
    g = util.random_genome(rng, 20000)
    t = util.make_text([g, util.mutate(rng, g, 0.001), np.full(3000, 2, dtype=np.uint8)])
    assert np.array_equal(host.build_bwt(t), oracle.bwt(t))


def test_sais_rejects_bad_text():
    with pytest.raises(ValueError):
        host.build_bwt(np.array([1, 2, 3], dtype=np.uint8))          # no final sentinel
    with pytest.raises(ValueError):
        host.build_bwt(np.array([1, 0, 0, 2, 0], dtype=np.uint8))    # empty string (out of contract)
    with pytest.raises(ValueError):
        host.build_bwt(np.array([1, 9, 0], dtype=np.uint8))


@pytest.mark.parametrize("name", [k for k, v in MAN.items() if "plain" in v])
def test_fmd_writer_bit_exact(name, tmp_path):
    ent = MAN[name]
    plain = gzip.open(os.path.join(util.GOLDEN, ent["plain"])).read().strip()
    fmd = host.fmd_bytes_from_plain(plain)   # the host FMD writer through the library
    assert hashlib.md5(fmd).hexdigest() == ent["fmd_md5"]
    assert fmd == open(os.path.join(util.GOLDEN, ent["fmd"]), "rb").read()
    src = tmp_path / "bwt.txt"                # ... and through the `plain2fmd` sub-command (the reference's main.c:299-331)
    src.write_bytes(plain + b"\n")
    assert run([CLI, "plain2fmd", str(src)]) != b""
    body = plain if plain.endswith(b"$") else plain
    src.write_bytes(body)
    assert run([CLI, "plain2fmd", str(src)]) == fmd
    out = tmp_path / "o.fmd"
    run([CLI, "plain2fmd", "-o", str(out), str(src)])
    assert out.read_bytes() == fmd


@pytest.mark.parametrize("name", ["genomes12", "reads_fq", "copies3000", "longruns", "edge_chars"])
def test_fmd_reader_and_recode(name, tmp_path):
    ent = MAN[name]
    src = os.path.join(util.GOLDEN, ent["fmd"])
    plain = run([CLI, "recode", src])
    assert hashlib.md5(plain).hexdigest() == ent["plain_md5"]
    assert run([CLI, "recode", "-d", src]) == open(src, "rb").read()
    fmr = tmp_path / "x.fmr"
    fmr.write_bytes(run([CLI, "recode", "-b", src]))
    assert run([CLI, "recode", "-d", str(fmr)]) == open(src, "rb").read()
    if os.path.exists(util.REF_BIN):  # our FMR must load in the reference
        assert run([util.REF_BIN, "build", "-i", str(fmr), "-d"]) == open(src, "rb").read()


def test_fmr_reader_on_reference_file():
    r = MAN["resume"]
    got = run([CLI, "recode", "-d", os.path.join(util.GOLDEN, r["fmr"])])
    assert got == open(os.path.join(util.GOLDEN, r["fmd"]), "rb").read()


def test_seqio_fasta_fastq_lines(tmp_path, oracle):
    fa = tmp_path / "a.fa"
    fa.write_bytes(b";junk before the first header\n>s1 comment\nACGT\nacgn\n\n>s2\nTT\r\n>s3\n>s4\nGRY*\n")
    got = list(host.read_batches(str(fa), False, 0))
    assert len(got) == 1 and got[0][0] == 6  # s3 is empty and skipped
    want = util.make_text([np.array([1, 2, 3, 4, 1, 2, 3, 5], dtype=np.uint8), np.array([4, 4], dtype=np.uint8), np.array([3, 5, 5, 5], dtype=np.uint8)])
    assert np.array_equal(got[0][1], want)
    fq = tmp_path / "a.fq.gz"
    with gzip.open(str(fq), "wb") as f:
        f.write(b"@r1\nACGT\n+\nIIII\n@r2 x\nGG\nCC\n+r2\nII\nII\n")
    got = list(host.read_batches(str(fq), False, 0, True, False))
    assert np.array_equal(got[0][1], np.array([1, 2, 3, 4, 0, 3, 3, 2, 2, 0], dtype=np.uint8))
    ln = tmp_path / "a.txt"
    ln.write_bytes(b"ACG\nTT\r\nA")
    got = list(host.read_batches(str(ln), True, 0, True, False))
    assert np.array_equal(got[0][1], np.array([1, 2, 3, 0, 4, 4, 0, 1, 0], dtype=np.uint8))
    # batching rule io.c:114,119: stop after the record that makes the batch longer than max_len
    got = list(host.read_batches(str(ln), True, 5, True, True))
    assert [g[0] for g in got] == [2, 2, 2] or [g[0] for g in got] == [4, 2]
    assert np.array_equal(np.concatenate([g[1] for g in got]), util.make_text([np.array([1, 2, 3], dtype=np.uint8), np.array([4, 4], dtype=np.uint8), np.array([1], dtype=np.uint8)]))


def _nt6_restated(line):
    """io.c:12-28 restated in numpy: A/C/G/T in either case -> 1..4, bytes 0..4 stay, everything else -> 5"""
    t = np.full(256, 5, dtype=np.uint8)
    t[:5] = np.arange(5)
    for ch, v in zip(b"ACGT", (1, 2, 3, 4)):
        t[ch] = t[ch + 32] = v
    return t[np.frombuffer(line, dtype=np.uint8)]


@pytest.mark.parametrize("fmt", ["lines", "fasta", "lines.gz"])
def test_seqio_vectorised_conversion_vs_restatement(tmp_path, fmt):
    """the reader converts 16 characters at a time (SSE2) and, for one-sequence-per-line input, straight out of its I/O buffer:
    random records of every length around the vector width and around the 1 MB buffer boundary, with lower case, IUPAC codes,
    arbitrary bytes, raw codes 0..4 (which io.c:12-28 leaves alone) and CRLF, against a numpy restatement of io.c:12-40,84-102
    (forward strand + reverse complement 1<->4, 2<->3, 0 and 5 unchanged)"""
    rng = np.random.default_rng(5)
    recs = []
    alph = np.frombuffer(b"ACGTacgtNnRYKM*-", dtype=np.uint8)
    for i in range(300):
        l = int(rng.choice([1, 2, 15, 16, 17, 31, 32, 33, 47, 48, 150, 151, 1000, 4099]))
        r = alph[rng.integers(0, len(alph), size=l)].copy()
        if i % 7 == 0:   # arbitrary bytes, raw codes among them (never a line feed; no '>' '@' '+' so that FASTA stays FASTA)
            k = rng.integers(0, l, size=max(1, l // 5))
            v = rng.integers(0, 256, size=len(k)).astype(np.uint8)
            v[np.isin(v, np.frombuffer(b"\n\r>@+", dtype=np.uint8))] = 1
            r[k] = v
        recs.append(r.tobytes())
    recs.append(bytes(alph[rng.integers(0, 8, size=(1 << 20) + 12345)]))   # longer than the reader's buffer
    recs += [bytes(alph[rng.integers(0, 8, size=70000)]) for _ in range(40)]  # so that lines straddle the buffer boundary
    path = tmp_path / ("in." + fmt)
    if fmt == "fasta":
        data = b"".join(b">r%d\n" % i + r[:len(r) // 2] + b"\n" + r[len(r) // 2:] + b"\r\n" for i, r in enumerate(recs))
    else:
        data = b"".join(r + (b"\r\n" if i % 3 == 0 and len(r) > 1 else b"\n") for i, r in enumerate(recs))
    if fmt.endswith(".gz"):
        with gzip.open(str(path), "wb") as f:
            f.write(data)
    else:
        path.write_bytes(data)
    comp = np.array([0, 4, 3, 2, 1, 5], dtype=np.uint8)
    want = []
    for r in recs:
        f = _nt6_restated(r)
        want += [f, np.zeros(1, dtype=np.uint8), comp[f[::-1]], np.zeros(1, dtype=np.uint8)]   # io.c:84-102: both strands, a sentinel each
    want = np.concatenate(want)
    got = list(host.read_batches(str(path), fmt != "fasta", 3000000))
    assert sum(g[0] for g in got) == 2 * len(recs)
    assert np.array_equal(np.concatenate([g[1] for g in got]), want)


_PIPE_READER = """
import sys, hashlib
sys.path.insert(0, %r)
from ropebwt3_amd import host
n, h = 0, hashlib.md5()
for k, t in host.read_batches(sys.argv[1], int(sys.argv[2]), 1 << 40):
    n += k
    h.update(t.tobytes())
print(n, h.hexdigest())
"""


@pytest.mark.parametrize("is_line", [0, 1])
@pytest.mark.parametrize("gz", [False, True])
def test_seqio_input_that_cannot_be_rewound(tmp_path, is_line, gz):
    """a pipe on stdin and a FIFO, plain and gzip-compressed, give the same batches as the file itself: the reader must not
    consume the bytes it sniffs (round 2 read the gzip magic off a pipe and then handed zlib a headerless stream, which it
    passed through as 'sequence')"""
    import sys
    rng = np.random.default_rng(5)
    recs = ["".join("ACGT"[x] for x in rng.integers(0, 4, size=int(rng.integers(30, 400)))) for _ in range(3000)]
    raw = ("\n".join(recs) + "\n").encode() if is_line else "".join(">r%d\n%s\n" % (i, r) for i, r in enumerate(recs)).encode()
    data = gzip.compress(raw) if gz else raw
    fn = tmp_path / ("in.gz" if gz else "in.txt")
    fn.write_bytes(data)
    script = _PIPE_READER % util.ROOT
    want = subprocess.run([sys.executable, "-c", script, str(fn), str(is_line)], stdout=subprocess.PIPE, check=True).stdout
    assert int(want.split()[0]) == 2 * len(recs)
    got = subprocess.run([sys.executable, "-c", script, "-", str(is_line)], input=data, stdout=subprocess.PIPE, check=True).stdout   # a pipe
    assert got == want
    fifo = str(tmp_path / "fifo")
    os.mkfifo(fifo)
    p = subprocess.Popen([sys.executable, "-c", script, fifo, str(is_line)], stdout=subprocess.PIPE)
    with open(fifo, "wb") as f:
        f.write(data)
    assert p.communicate(timeout=120)[0] == want and p.returncode == 0
    with open(fn, "rb") as f:   # stdin redirected from the file itself (seekable: the fast path)
        assert subprocess.run([sys.executable, "-c", script, "-", str(is_line)], stdin=f, stdout=subprocess.PIPE, check=True).stdout == want


def test_seqio_truncated_gzip_is_an_error_not_an_end_of_file(tmp_path):
    recs = ["ACGT" * 50] * 4000
    data = gzip.compress(("\n".join(recs) + "\n").encode())
    fn = tmp_path / "cut.gz"
    fn.write_bytes(data[:len(data) // 2])
    with pytest.raises(ValueError):
        list(host.read_batches(str(fn), True, 1 << 40))


def test_parse_num():
    assert host.parse_num("7g") == 7000000000 and host.parse_num("500k") == 500000 and host.parse_num("2.5M") == 2500000 and host.parse_num("13") == 13


def test_c_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(_build.INCLUDE, "rb3gpu.h")).read()
    declared = set(re.findall(r"\b(rb3gpu_[a-z0-9_]+)\s*\(", hdr)) - {"rb3gpu_emit_f"}
    assert declared == set(gpu.SYMBOLS), (declared ^ set(gpu.SYMBOLS))
    lib = gpu.load_library()  # raises if the .so is missing; no compute call is made here
    for name in declared:
        assert hasattr(lib, name)
    assert lib.rb3gpu_strerror(-4).decode().startswith("BWT symbol")


def test_no_cpu_fallback_without_device():
    lib = gpu.load_library()
    if lib.rb3gpu_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError):
        gpu.Rb3Gpu()
    r = subprocess.run([CLI, "build", "-L", os.path.join(util.GOLDEN, "k2_fwd.txt")], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode != 0 and b"no CPU fallback" in r.stderr


def test_walkers_from_sampled_isa_equal_the_sorter_walkers():
    """rb3h_walkers_from_ckrow (walker list from a sampled inverse suffix array, as the GPU sorter returns it)
    gives the list rb3h_build_bwt_walkers derives from its own suffix array"""
    from ropebwt3_amd import host
    rng = np.random.default_rng(5)
    seqs = [util.random_genome(rng, 5000), util.random_genome(rng, 130), util.random_genome(rng, 900)]
    t = util.make_text(seqs)
    n = t.size
    sid = np.cumsum(t == 0) - (t == 0)

    def key(p):
        e = p
        while t[e] != 0:
            e += 1
        return (bytes(t[p:e]), sid[p])
    sa = sorted(range(n), key=key)
    isa = np.empty(n, dtype=np.int64)
    isa[sa] = np.arange(n)
    for step in (64, 100, 384):
        _, w = host.build_bwt_walkers(t, step)
        assert np.array_equal(w, host.walkers_from_ckrow(t, step, isa[::step].copy()))


@pytest.mark.parametrize("case", ["w32", "w64", "mixed"])
def test_fmd_writer_wide_block_headers_vs_reference_library(tmp_path, case):
    """FMD blocks whose predecessor holds >= 2^14 symbols carry 32-bit headers and >= 2^30 symbols 64-bit headers
    (rld0.c:116-128).  Runs that long are fed to the reference's own encoder (rld_init / rld_enc / rld_enc_finish / rld_dump
    of oracle/_ref/librb3ref.so -- no text of that size is needed) and to the host writer (rb3h_fmdw_*): the files must be
    byte-identical.  This is the only place 64-bit headers are exercised."""
    import ctypes
    from ropebwt3_amd import host
    if not util.Reference.available():
        pytest.skip("no reference library")
    R = util.Reference().L
    R.rld_init.restype = ctypes.c_void_p
    R.rld_init.argtypes = [ctypes.c_int, ctypes.c_int]
    R.rld_itr_init.restype = None
    R.rld_itr_init.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
    R.rld_enc.restype = ctypes.c_int
    R.rld_enc.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint8]
    R.rld_enc_finish.restype = ctypes.c_uint64
    R.rld_enc_finish.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    R.rld_dump.restype = ctypes.c_int
    R.rld_dump.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
    R.rld_destroy.restype = None
    R.rld_destroy.argtypes = [ctypes.c_void_p]
    rng = np.random.default_rng({"w32": 1, "w64": 2, "mixed": 3}[case])
    runs = []
    for i in range(400):
        if case == "w32":
            l = int(rng.choice([1, 3, 17, 300, 20000, 70000, 1 << 20, (1 << 29) + 5]))
        elif case == "w64":
            l = int(rng.choice([1, 2, (1 << 30) + 3, (1 << 31) + 1, 1 << 33, (1 << 40) + 12345, 9]))
        else:
            l = int(rng.choice([1, 1, 2, 5, 40, 1000, 16384, 16383, (1 << 30) - 1, 1 << 30, (1 << 30) + 1, 1 << 36]))
        c = int(rng.integers(0, 6))
        if runs and runs[-1][1] == c:
            c = (c + 1) % 6
        runs.append((l, c))
    e = R.rld_init(6, 3)
    itr = ctypes.create_string_buffer(256)
    R.rld_itr_init(e, itr, 0)
    for l, c in runs:
        R.rld_enc(e, itr, l, c)
    R.rld_enc_finish(e, itr)
    fn_ref = str(tmp_path / "ref.fmd")
    assert R.rld_dump(e, fn_ref.encode()) == 0
    R.rld_destroy(e)
    H = host.load_library()
    H.rb3h_fmdw_dump_file.restype = ctypes.c_int
    H.rb3h_fmdw_dump_file.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
    w = H.rb3h_fmdw_init()
    for l, c in runs:
        assert H.rb3h_fmdw_enc(w, l, c) == 0
    assert H.rb3h_fmdw_finish(w) == 0
    fn_own = str(tmp_path / "own.fmd")
    assert H.rb3h_fmdw_dump_file(w, fn_own.encode()) == 0
    H.rb3h_fmdw_destroy(w)
    a, b = open(fn_ref, "rb").read(), open(fn_own, "rb").read()
    assert len(a) == len(b) and a == b
    # and the reader gets the runs back
    out = subprocess.run([os.path.join(os.path.dirname(host.__file__), "ropebwt3-amd"), "recode", "-d", fn_own], stdout=subprocess.PIPE)
    assert out.returncode == 0 and out.stdout == a


def test_walker_list_by_text_position_covers_every_row_once():
    """rb3h_walkers_text: per string one walker at the sentinel and one at every multiple of `step` strictly inside it, in text
    order; an inner walker starts `flags >> 8` positions (16 = the age from which the engine's walkers record; fewer next to
    the string's end) to the RIGHT of its segment and its nsteps counts them; the segments of a string tile it exactly; the
    sentinel walker's own segment is never shorter than 128 steps where the string is that long"""
    PRE = 16   # RB3H_PREROLL (host/sais.c) = RB3_TENT_MIN_AGE of the engine
    rng = np.random.default_rng(77)
    seqs = [util.random_genome(rng, n) for n in (5000, 1, 257, 384, 385, 20000, 130, 3)]
    t = util.make_text(seqs, True, False)
    ends = np.flatnonzero(t == 0)
    for step in (192, 220, 384, 1000):
        w = host.walkers_text(t, step)
        assert w.shape[1] == 4
        i = 0
        b = 0
        for e in ends:                                    # string [b, e), sentinel at e
            inner = []
            while w[i, 1] != -2:                          # inner walkers of this string come first, left to right
                inner.append(w[i]); i += 1
            sent = w[i]; i += 1
            assert sent[0] == e and sent[3] == 0
            prev = None
            for row, ka0, nsteps, flags in inner:
                pre, probe = flags >> 8 & 0xFF, flags >> 16
                p = row - pre
                assert ka0 == -1 and flags & 0xFF == 0 and 0 <= pre <= PRE and p % step == 0 and b < p < e and row < e
                assert probe == (64 if pre == PRE and e - 1 - row >= 64 else 0) and row + probe < e
                assert pre == min(PRE, e - 1 - p)
                assert nsteps == (np.iinfo(np.int64).max // 2 if prev is None else p - prev + pre)
                prev = p
            assert sent[2] == (np.iinfo(np.int64).max // 2 if prev is None else e - prev)
            if prev is not None:
                assert e - prev >= min(128, step)
            b = e + 1
        assert i == w.shape[0]


def test_parallel_host_sorter_equals_sais(monkeypatch):
    """rb3h_build_bwt / rb3h_build_bwt_walkers with several threads (psort.c: prefix doubling over OpenMP, what rb3_build_sais gets from
    libsais + OpenMP, sais-ss.c:15-22) give the bytes and the walker list of the sequential SA-IS: a genome on both strands, reads,
    relatives, exact copies in one batch (the parallel sorter gives up on long repeats and SA-IS takes over), homopolymers, ragged strings"""
    from ropebwt3_amd import host
    monkeypatch.setenv("RB3H_PSORT_MIN_THREADS", "2")
    rng = np.random.default_rng(7)
    g = util.random_genome(rng, 150000)
    cases = [util.make_text([g]),
             util.make_text(util.reads_from(rng, g, 2000, 100, err=0.01)),
             util.make_text([util.mutate(rng, g[:30000], 0.002) for _ in range(6)]),
             util.make_text([g[:40000].copy() for _ in range(4)], rev=False),
             util.make_text([np.full(70000, 1, dtype=np.uint8), np.full(3000, 2, dtype=np.uint8), g[:2000]], rev=False),
             util.make_text([g[:100000], g[5:6].copy(), g[7:300]] + util.reads_from(rng, g, 300, 33))]
    for t in cases:
        assert t.size >= 1 << 16
        assert np.array_equal(host.build_bwt(t, 1), host.build_bwt(t, 3))
        wa, wb = host.build_bwt_walkers(t, 512, 1), host.build_bwt_walkers(t, 512, 4)
        assert np.array_equal(wa[0], wb[0]) and np.array_equal(wa[1], wb[1])
    monkeypatch.setenv("RB3H_PSORT_FORCE64", "1")   # the instantiation for batches of 2^32 symbols and more, on a small one
    for t in cases[:3]:
        assert np.array_equal(host.build_bwt(t, 1), host.build_bwt(t, 4))
    monkeypatch.delenv("RB3H_PSORT_FORCE64")
    monkeypatch.setenv("RB3H_PSORT_MEM_LIMIT", "1000000")   # less memory than the parallel sorter needs: it declines BEFORE allocating, SA-IS takes the batch
    assert np.array_equal(host.build_bwt(cases[0], 1), host.build_bwt(cases[0], 4))


def test_bench_self_launch_prints_exactly_one_json_line():
    """`python bench.py --gpus 2` WITHOUT a launcher: the parent starts the ranks itself (torch.distributed.run); rank 0's
    JSON line must arrive on the parent's stdout -- exactly one line, nothing else -- whatever libraries print (the ranks redirect their own
    descriptor 1; the parent's saved descriptor number must not leak into them).  RB3_BENCH_LAUNCH_SELFTEST stops the ranks before they touch a GPU."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RB3_BENCH_LAUNCH_SELFTEST="1")
    env.pop("RANK", None), env.pop("WORLD_SIZE", None), env.pop("LOCAL_RANK", None)
    env["RB3_BENCH_STDOUT_FD"] = "987"   # a stale descriptor number in the environment (what round 5's parent leaked) must not matter either
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, (r.stdout, r.stderr[-2000:])
    d = json.loads(lines[0])
    assert d["selftest"] == "launch" and d["world"] == 2 and d["n_gpus"] == 2
    assert "noise a library would write" in r.stderr and "noise" not in r.stdout
