"""The byte-range reader behind `build --gpus N` on one file (rb3h_seq_open_range): ranges that tile a file give every record to
exactly one reader, in file order -- for one-sequence-per-line input, FASTA (also multi-line) and FASTQ whose quality lines
begin with '@'.  CPU only (the host library)."""
import os
import random

import numpy as np
import pytest

from ropebwt3_amd import host


def _read_all(fn, is_line, ranges):
    out = b""
    for beg, end in ranges:
        for _, t in host.read_batches(fn, is_line, 1 << 40, byte_range=None if (beg, end) == (0, 0) else (beg, end)):
            out += t.tobytes()
    return out


@pytest.mark.parametrize("kind", ["line", "fasta", "fasta_multiline", "fastq", "fasta_blank_first", "fasta_at_headers", "fastq_multiline",
                                  "fastq_crlf", "fastq_no_final_newline", "garbage_first"])
def test_ranges_tile_a_file(tmp_path, kind):
    rng = random.Random(5)
    seqs = ["".join(rng.choice("ACGT") for _ in range(rng.randint(1, 300))) for _ in range(400)]
    fn = str(tmp_path / "x")
    with open(fn, "w") as f:
        for i, s in enumerate(seqs):
            if kind == "line":
                f.write(s + "\n")
            elif kind in ("fasta", "fasta_blank_first", "garbage_first"):
                if i == 0 and kind != "fasta":   # (files the cut detector does not vouch for must still tile)
                    f.write("\n\n" if kind == "fasta_blank_first" else "x\n")
                f.write(">s%d\n%s\n" % (i, s))
            elif kind == "fasta_at_headers":      # kseq takes '@' for a header in a FASTA file too
                f.write("%ss%d\n%s\n" % (">" if i % 3 else "@", i, s) if i else ">s0\n%s\n" % s)
            elif kind == "fastq_multiline":
                q = "".join(rng.choice("@+>I#") for _ in s)
                f.write("@s%d\n" % i + "\n".join(s[k:k + 50] for k in range(0, len(s), 50)) + "\n+\n" + "\n".join(q[k:k + 50] for k in range(0, len(q), 50)) + "\n")
            elif kind in ("fastq_crlf", "fastq_no_final_newline"):
                q = "".join(rng.choice("@+>I#") for _ in s)
                nl = "\r\n" if kind == "fastq_crlf" else "\n"
                rec = "@s%d%s%s%s+%s%s%s" % (i, nl, s, nl, nl, q, nl)
                f.write(rec[:-1] if kind == "fastq_no_final_newline" and i + 1 == len(seqs) else rec)
            elif kind == "fasta_multiline":
                f.write(">s%d\n" % i + "\n".join(s[k:k + 60] for k in range(0, len(s), 60)) + "\n")
            else:
                q = "".join(rng.choice("@+>I#") for _ in s)
                q = ("@" + q[1:]) if i % 2 == 0 else q
                f.write("@s%d\n%s\n+\n%s\n" % (i, s, q))
    size = os.path.getsize(fn)
    whole = _read_all(fn, kind == "line", [(0, 0)])
    assert whole.count(b"\0") == 2 * len(seqs)
    for n in (2, 3, 5, 17, 64, 401):
        cuts = [size * k // n for k in range(n + 1)]
        parts = _read_all(fn, kind == "line", [(cuts[k], cuts[k + 1] if k + 1 < n else 0) for k in range(n)])
        assert parts == whole, (kind, n)
    # arbitrary cut points, including ones just before / on / after record boundaries
    for _ in range(30):
        c = sorted(rng.sample(range(1, size), 4))
        cuts = [0] + c + [size]
        parts = _read_all(fn, kind == "line", [(cuts[k], cuts[k + 1] if k + 2 < len(cuts) else 0) for k in range(len(cuts) - 1)])
        assert parts == whole, (kind, cuts)
