"""The native FASTA/FASTQ(.gz) batch decoder (csrc/tps_io.cpp) yields exactly the records of the
pure-Python parser, in batches that respect the size limits."""
import gzip
import os

import numpy as np
import pytest

from topsicle_amd import seqio


def all_records(path, **kw):
    out = []
    for b in seqio.read_batches(path, **kw):
        out += [b.record(i) for i in range(len(b))]
    return out


@pytest.mark.skipif(seqio._load_io() is None, reason="libtopsicle_io.so not built")
def test_native_equals_python_on_demo(gold_dir):
    path = os.path.join(gold_dir, "demo_col0.fastq.gz")
    py = list(seqio.read_records(path))
    nat = all_records(path)
    assert len(nat) == len(py) == 44
    for a, b in zip(nat, py):
        assert (a.id, a.description, a.seq, a.qual) == (b.id, b.description, b.seq, b.qual)
    small = all_records(path, max_bases=60000)          # many small batches, same records
    assert [r.seq for r in small] == [r.seq for r in py]
    sizes = [int(b.offsets[-1]) for b in seqio.read_batches(path, max_bases=60000)]
    assert len(sizes) > 10 and all(s <= 60000 for s in sizes)


@pytest.mark.skipif(seqio._load_io() is None, reason="libtopsicle_io.so not built")
def test_native_fasta_multiline_crlf_and_edge_cases(tmp_path):
    fa = tmp_path / "x.fasta"
    fa.write_bytes(b">r1 first read\r\nACGT\r\nacgtn\r\n\r\n>r2\nTTTT\n>r3 empty\n>r4\nGG GG\n")
    recs = all_records(str(fa))
    assert [(r.id, r.description, r.seq) for r in recs] == [("r1", "r1 first read", "ACGTacgtn"), ("r2", "r2", "TTTT"),
                                                            ("r3", "r3 empty", ""), ("r4", "r4", "GG GG")]
    assert [(r.id, r.seq) for r in seqio.read_records(str(fa))] == [(r.id, r.seq) for r in recs]
    fq = tmp_path / "y.fq.gz"
    with gzip.open(fq, "wt") as h:
        h.write("@a x\nACGT\n+\nIIII\n@b\nAC\nGT\n+b\nII\nII\n@c\n\n+\n\n@d\nA\n+\n#\n")
    recs = all_records(str(fq))
    assert [(r.id, r.seq, r.qual) for r in recs] == [("a", "ACGT", "IIII"), ("b", "ACGT", "IIII"), ("c", "", ""), ("d", "A", "#")]
    empty = tmp_path / "empty.fastq"
    empty.write_bytes(b"")
    assert all_records(str(empty)) == []
    big = tmp_path / "big.fasta"
    big.write_text(">long\n" + "ACGT" * 5000 + "\n")
    recs = all_records(str(big), max_bases=1000)        # a single record larger than the batch grows the buffers
    assert len(recs) == 1 and len(recs[0].seq) == 20000


@pytest.mark.skipif(seqio._load_io() is None, reason="libtopsicle_io.so not built")
@pytest.mark.parametrize("threads", ["1", "5"])
def test_native_plain_fastq_thread_team(tmp_path, monkeypatch, threads):
    """Plain 4-line FASTQ goes through the mmap + thread-team decoder; anything irregular (wrapped lines,
    blank lines, length mismatch) must hand the rest of the file to the streaming decoder with no record
    lost or duplicated."""
    monkeypatch.setenv("TPS_IO_THREADS", threads)
    rng = np.random.default_rng(11)

    def rec(i, n, crlf=False):
        s = "".join("ACGTNacgt"[x] for x in rng.integers(0, 9, n))
        q = "".join(chr(33 + int(x)) for x in rng.integers(0, 40, n))
        nl = "\r\n" if crlf else "\n"
        return f"@read{i} desc {i}{nl}{s}{nl}+{nl}{q}{nl}", (f"read{i}", f"read{i} desc {i}", s, q)

    # (a) regular file, no trailing newline on the last line, several batches and a re-indexed window
    parts, want = zip(*[rec(i, int(rng.integers(0, 3000)), crlf=(i % 7 == 0)) for i in range(400)])
    p = tmp_path / "plain.fastq"
    p.write_text("".join(parts).rstrip("\n"), newline="")
    for mb in (1 << 20, 20000, 3500):
        got = all_records(str(p), max_bases=mb)
        assert [(r.id, r.description, r.seq, r.qual) for r in got] == list(want), mb
    assert [(r.id, r.seq, r.qual) for r in seqio.read_records(str(p))] == [(w[0], w[2], w[3]) for w in want]
    # (b) irregular records in the middle: wrapped sequence, blank line, '+name' separator line
    irregular = "@w1\nACGT\nACGT\n+\nIIII\nIIII\n\n@w2 x\nAAAA\n+w2\nIIII\n"
    q = tmp_path / "mixed.fastq"
    q.write_text("".join(parts[:50]) + irregular + "".join(parts[50:60]) + "\n\n", newline="")
    got = all_records(str(q), max_bases=30000)
    exp = list(want[:50]) + [("w1", "w1", "ACGTACGT", "IIIIIIII"), ("w2", "w2 x", "AAAA", "IIII")] + list(want[50:60])
    assert [(r.id, r.description, r.seq, r.qual) for r in got] == exp
    # (c) a record larger than the batch buffers grows them; leading blank lines are skipped
    big = tmp_path / "big.fastq"
    big.write_text("\n\n@big\n" + "ACGT" * 5000 + "\n+\n" + "I" * 20000 + "\n@s\nAC\n+\nII\n")
    got = all_records(str(big), max_bases=1000)
    assert [(r.id, len(r.seq)) for r in got] == [("big", 20000), ("s", 2)]


def _write_bgzf(path, payload: bytes, block=60000, splits=None):
    """bgzip-compatible file: independent gzip members with the 'BC' extra field (block size - 1), then the empty
    end-of-file block."""
    import struct
    import zlib

    def member(chunk):
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = co.compress(chunk) + co.flush()
        bsize = 12 + 6 + len(body) + 8
        return (b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1) +
                body + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
    cuts = splits or list(range(block, len(payload), block))
    with open(path, "wb") as h:
        lo = 0
        for hi in cuts + [len(payload)]:
            h.write(member(payload[lo:hi]))
            lo = hi
        h.write(member(b""))


@pytest.mark.skipif(seqio._load_io() is None, reason="libtopsicle_io.so not built")
@pytest.mark.parametrize("threads", ["1", "6"])
def test_native_bgzf_blocks_inflate_in_parallel(tmp_path, monkeypatch, caplog, threads):
    """bgzip'ed FASTQ: the blocks are inflated by the thread team and decoded like plain FASTQ; records span block
    boundaries; an irregular record hands over to the streaming decoder; a corrupt block is an error, not a short file."""
    monkeypatch.setenv("TPS_IO_THREADS", threads)
    rng = np.random.default_rng(11)
    recs = []
    for i in range(900):
        L = int(rng.integers(0, 9000))
        seq = "".join("ACGTN"[x] for x in rng.integers(0, 5, L))
        qual = "".join(chr(33 + int(x)) for x in rng.integers(0, 60, L))
        recs.append((f"read{i} desc {i}", seq, qual))
    text = "".join(f"@{h}\n{s}\n+\n{q}\n" for h, s, q in recs)
    p = tmp_path / "reads.fastq.gz"
    _write_bgzf(str(p), text.encode(), block=50021)
    got = all_records(str(p), max_bases=700000)
    assert [(r.description, r.seq, r.qual) for r in got] == recs
    monkeypatch.setenv("TPS_IO_BGZF_GROUP", "300000")       # many refills: partial records are carried over
    got = all_records(str(p), max_bases=2000000)
    assert [(r.description, r.seq, r.qual) for r in got] == recs
    monkeypatch.delenv("TPS_IO_BGZF_GROUP")
    assert [(r.description, r.seq) for r in seqio.read_records(str(p))] == [(h, s) for h, s, _ in recs]     # it is plain gzip too
    # wrapped sequence lines in the middle: the streaming decoder takes over at that record
    odd = text + "@wrapped\nAC\nGT\n+\nII\nII\n@last\nA\n+\n#\n"
    p2 = tmp_path / "odd.fq.gz"
    _write_bgzf(str(p2), odd.encode(), block=65000)
    got = all_records(str(p2))
    assert [(r.id, r.seq, r.qual) for r in got[-2:]] == [("wrapped", "ACGT", "IIII"), ("last", "A", "#")] and len(got) == len(recs) + 2
    # corrupt payload
    raw = bytearray(open(p, "rb").read())
    raw[len(raw) // 2] ^= 0xFF
    p3 = tmp_path / "bad.fastq.gz"
    p3.write_bytes(bytes(raw))
    import logging
    caplog.set_level(logging.ERROR)
    assert all_records(str(p3)) == []                   # errors are logged, nothing is returned (the reference's convention)
    assert any("Error parsing file" in r.getMessage() for r in caplog.records)
