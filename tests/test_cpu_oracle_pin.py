"""Pins the CPU oracle (oracle/oracle.c) to the reference: golden vectors produced by the
unmodified reference binary (tests/golden, tools/make_golden.py) and, when oracle/_ref exists,
the reference's own shared object on fresh random inputs."""
import gzip
import json
import os

import numpy as np
import pytest

from tests import util

MAN = json.load(open(os.path.join(util.GOLDEN, "MANIFEST.json")))
SYM2CODE = {ord(c): i for i, c in enumerate(util.SYMS)}


def golden_plain(name):
    raw = gzip.open(os.path.join(util.GOLDEN, MAN[name]["plain"])).read().strip()
    return np.array([SYM2CODE[c] for c in raw], dtype=np.uint8)


def read_lines(name):
    fn = os.path.join(util.GOLDEN, MAN[name]["inputs"][0])
    raw = gzip.open(fn).read() if fn.endswith(".gz") else open(fn, "rb").read()
    lines = raw.decode("latin1").split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    return [l[:-1] if len(l) > 1 and l.endswith("\r") else l for l in lines]


def test_known_answers(oracle):
    # K1/K2, K3, K4 and the duplicates/N/lower-case vector
    assert util.sym_str(oracle.bwt(oracle.text(["AGG", "AGC"], True, False))) == "GC$$GGAA"
    assert util.sym_str(oracle.bwt(oracle.text(["AGG", "AGC"]))) == "GTCT$$G$CGGA$ACC"
    assert util.sym_str(oracle.bwt(oracle.text(["TGAACTCTACACAACATATTTTGTCACCAAG"]))) == \
        "GACCACGCAGCTACAATGACTTTAACATAATA$ATTATTTGTATCGAATGC$GTGTTAGCTGTA"
    assert util.sym_str(oracle.bwt(oracle.text(["ACG", "ACG", "TTT", "ACG", "NNA", "acgtn"]))) == \
        "GTGTTAGTANNTANA$$$$N$AAA$$$AACCCCCCCCGGTGGT$G$NTN$T$"
    for name in ("k2_fwd", "k3_both", "k4_readme", "edge_dups"):
        assert MAN[name]["plain_text"] == {"k2_fwd": "GC$$GGAA", "k3_both": "GTCT$$G$CGGA$ACC"}.get(name, MAN[name]["plain_text"])


def test_k2_merge_trace(oracle):
    # B1 = G$GA, B2 = C$GA; chain rows 0->2->3->1 land on merged positions 1,4,6,2
    b1 = oracle.bwt(oracle.text(["AGG"], True, False))
    b2 = oracle.bwt(oracle.text(["AGC"], True, False))
    assert util.sym_str(b1) == "G$GA" and util.sym_str(b2) == "C$GA"
    rb, acc2 = oracle.mg_rank(b1, b2)
    assert list(rb >> 6) == [1, 2, 4, 6]
    assert list(rb >> 3 & 7) == list(b2)
    assert list(acc2) == [0, 1, 2, 3, 4, 4, 4]
    assert util.sym_str(oracle.merge(b1, b2)) == "GC$$GGAA"


@pytest.mark.parametrize("name,fwd,rev", [("edge_dups", True, True), ("edge_chars", True, True), ("reads_fwd", True, False),
                                          ("k2_fwd", True, False), ("k3_both", True, True)])
def test_line_fixtures_one_batch_and_merged(oracle, name, fwd, rev):
    want = golden_plain(name)
    lines = read_lines(name)
    assert np.array_equal(oracle.bwt(oracle.text(lines, fwd, rev)), want)
    # the same through the merge path, one record per batch up to a cap, then the rest in one go
    cur = oracle.bwt(oracle.text(lines[:1], fwd, rev))
    cut = min(len(lines), 40)
    for l in lines[1:cut]:
        cur = oracle.merge(cur, oracle.bwt(oracle.text([l], fwd, rev)))
    if cut < len(lines):
        cur = oracle.merge(cur, oracle.bwt(oracle.text(lines[cut:], fwd, rev)))
    assert np.array_equal(cur, want)


def _fasta_records(raw):
    recs, cur = [], None
    for l in raw.decode().split("\n"):
        if l.startswith(">"):
            if cur is not None:
                recs.append("".join(cur))
            cur = []
        elif cur is not None:
            cur.append(l)
    if cur is not None:
        recs.append("".join(cur))
    return recs


def test_genomes_fixture_merge_rounds(oracle):
    want = golden_plain("genomes12")
    recs = _fasta_records(gzip.open(os.path.join(util.GOLDEN, "genomes12.fa.gz")).read())
    assert len(recs) == 12
    cur = oracle.bwt(oracle.text(recs[:3]))
    for i in range(3, 12, 3):
        cur = oracle.merge(cur, oracle.bwt(oracle.text(recs[i:i + 3])))
    assert np.array_equal(cur, want)


@pytest.mark.skipif(not util.Reference.available(), reason="oracle/_ref not built (no /root/reference here)")
def test_against_reference_library_random(oracle):
    ref = util.Reference()
    rng = np.random.default_rng(99)
    for it in range(20):
        g = util.random_genome(rng, int(rng.integers(50, 3000)))
        s1 = [util.mutate(rng, g, 0.02) for _ in range(int(rng.integers(1, 4)))]
        s2 = util.reads_from(rng, g, int(rng.integers(1, 30)), min(40, len(g))) + [s1[0]]
        t1, t2 = util.make_text(s1), util.make_text(s2)
        b1, b2 = oracle.bwt(t1), oracle.bwt(t2)
        assert np.array_equal(b1, ref.bwt(t1)) and np.array_equal(b2, ref.bwt(t2))
        # reference merge: plain2fmr + merge_plain, then print
        assert np.array_equal(oracle.merge(b1, b2), ref.bwt(np.concatenate([t1, t2])))


@pytest.mark.parametrize("name", ["k2_fwd", "k3_both", "k4_readme", "edge_dups", "edge_chars", "genomes12", "reads_fq", "reads_fwd"])
def test_ssa_matches_reference_files(oracle, name):
    """sampled suffix array: the oracle's restatement of rb3_ssa_gen reproduces the .ssa
    files the reference's `ssa` wrote for the fixtures' indexes, byte for byte"""
    import hashlib
    b = golden_plain(name)
    for ss, md5 in MAN[name]["ssa_md5"].items():
        ms, r2i, ssa = oracle.ssa_gen(b, int(ss))
        out = util.ssa_bytes(int(ss), ms, r2i, ssa)
        assert hashlib.md5(out).hexdigest() == md5, (name, ss)
    if "ssa_file" in MAN[name]:
        f = MAN[name]["ssa_file"]
        ms, r2i, ssa = oracle.ssa_gen(b, f["shift"])
        assert util.ssa_bytes(f["shift"], ms, r2i, ssa) == open(os.path.join(util.GOLDEN, f["file"]), "rb").read()


def _rb_cases():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden_rb", os.path.join(util.ROOT, "tools", "make_golden_rb.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.cases()


@pytest.mark.parametrize("name", ["k2", "k3", "family", "reads"])
def test_rb_vectors_of_the_reference(oracle, name):
    """rb[] as the UNMODIFIED reference's rb3_mg_rank_plain left it (tests/golden/rb_vectors.npz, made by
    tools/make_golden_rb.py from oracle/_ref/librb3ref.so) -- every bit of every word equals the oracle's restatement of
    fm-index.c:160-175, 202-225: merged position, inserted symbol and bucket symbol"""
    vec = np.load(os.path.join(util.GOLDEN, "rb_vectors.npz"))
    l1, l2, both = _rb_cases()[name]
    b1, b2 = oracle.bwt(oracle.text(l1, True, both)), oracle.bwt(oracle.text(l2, True, both))
    rb, acc2 = oracle.mg_rank(b1, b2)
    assert np.array_equal(rb, vec[name + "_rb"]) and np.array_equal(acc2, vec[name + "_acc2"])
    if util.Reference.available():   # and live, where the reference's shared object travelled with the repository
        ref = util.Reference()
        rb2, acc3 = ref.mg_rank(b1, b2)
        assert np.array_equal(rb2, rb) and np.array_equal(acc3, acc2)


@pytest.mark.skipif(not util.Reference.available(), reason="oracle/_ref/librb3ref.so not built")
def test_rb_against_the_reference_on_random_inputs(oracle):
    ref = util.Reference()
    rng = np.random.default_rng(77)
    for _ in range(6):
        g0 = util.random_genome(rng, int(rng.integers(200, 4000)))
        old = [g0] + [util.mutate(rng, g0, 0.01) for _ in range(int(rng.integers(0, 4)))]
        new = [util.mutate(rng, g0, 0.01) for _ in range(int(rng.integers(1, 4)))] + util.reads_from(rng, g0, 5, 40)
        b1, b2 = oracle.bwt(util.make_text(old)), oracle.bwt(util.make_text(new))
        a, b = oracle.mg_rank(b1, b2), ref.mg_rank(b1, b2)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
