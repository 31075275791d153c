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
    seqio.io_option("threads", int(threads))
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
    # (c) a record larger than the batch buffers grows them
    big = tmp_path / "big.fastq"
    big.write_text("@big\n" + "ACGT" * 5000 + "\n+\n" + "I" * 20000 + "\n@s\nAC\n+\nII\n")
    got = all_records(str(big), max_bases=1000)
    assert [(r.id, len(r.seq)) for r in got] == [("big", 20000), ("s", 2)]
    # (d) both readers agree on odd input: a file that starts with a blank line has no identifiable format
    # (check_file_type, allsteps.py:36-50), and a quality string of the wrong length is an error, not a silent mis-framing
    blank = tmp_path / "blank.fastq"
    blank.write_text("\n@a\nACGT\n+\nIIII\n")
    assert all_records(str(blank)) == [] and list(seqio.read_records(str(blank))) == []
    short = tmp_path / "short.fastq"
    short.write_text("@a\nACGT\n+\nIIII\n@b\nACGTA\n+\nIIII\n@c\nAC\n+\nII\n")
    assert [r.id for r in all_records(str(short))] == ["a"]                 # the native reader stops at the bad record ...
    assert [r.id for r in seqio.read_records(str(short))] == ["a"]          # ... and so does the Python parser (both log the error)
    # a quality line shorter than the sequence whose missing characters are exactly the next header + newline: the packed reader
    # skips quality lines by length, but scans their bytes when what follows the record is not a record start -- the case here --
    # and rejects the record itself, like the streaming decoder (round 2 accepted "b" and failed one record later)
    late = tmp_path / "late.fastq"
    late.write_text("@a\nACGT\n+\nIIII\n@b\nACGTACGTAC\n+\nIIII\n@cccc\nACGTAC\n+\nIIIIII\n@d\nAC\n+\nII\n")
    pool = seqio.BufferPool(2, 4096, 64)
    ids = [pb.read_id(i) for pb in seqio.read_batches_packed(str(late), pool) for i in range(pb.n)]
    assert ids == ["a"] and [r.id for r in seqio.read_records(str(late))] == ["a"]


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
    seqio.io_option("threads", int(threads))
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
    seqio.io_option("bgzf_group", 300000)       # many refills: partial records are carried over
    got = all_records(str(p), max_bases=2000000)
    assert [(r.description, r.seq, r.qual) for r in got] == recs
    seqio.io_option("bgzf_group", 0)
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


# ---------------------------------------------------------------------------- packed batches (the upload format)
def _packed_records(path, words_cap=1 << 20, reads_cap=4096, n_sets=3):
    """(records, flat packed arrays re-packed per batch) through read_batches_packed."""
    from packfmt import np_pack
    pool = seqio.BufferPool(n_sets, words_cap, reads_cap)
    recs, n_batches = [], 0
    for pb in seqio.read_batches_packed(path, pool, max_records=reads_cap):
        rb = [pb.record(i) for i in range(pb.n)]
        bases = np.frombuffer("".join(r.seq for r in rb).encode("latin1"), np.uint8)
        offsets = np.zeros(pb.n + 1, np.int64)
        np.cumsum([len(r.seq) for r in rb], out=offsets[1:])
        seq2, inv, desc = np_pack(bases, offsets)
        assert np.array_equal(pb.seq2, seq2) and np.array_equal(pb.inv, inv) and np.array_equal(pb.desc, desc)
        assert pb.any_invalid == bool(inv.any())
        pb.release()
        assert [pb.record(i).seq for i in range(pb.n)] == [r.seq for r in rb]      # still readable after the buffers went back
        recs += rb
        n_batches += 1
    return recs, n_batches


@pytest.mark.skipif(seqio._load_io() is None, reason="libtopsicle_io.so not built")
def test_packed_reader_equals_python_parser(tmp_path, gold_dir):
    demo = os.path.join(gold_dir, "demo_col0.fastq.gz")
    py = list(seqio.read_records(demo))
    plain = tmp_path / "demo.fastq"
    with gzip.open(demo, "rb") as g:
        plain.write_bytes(g.read())
    crlf = tmp_path / "crlf.fastq"
    crlf.write_bytes(plain.read_bytes().replace(b"\n", b"\r\n"))
    for path, min_batches in ((str(plain), 1), (demo, 1), (str(crlf), 1)):
        recs, _ = _packed_records(path)
        assert [(r.id, r.description, r.seq, r.qual) for r in recs] == [(r.id, r.description, r.seq, r.qual) for r in py], path
    recs, nb = _packed_records(str(plain), words_cap=8192, reads_cap=16)          # many small batches (<= 131 kb each)
    assert nb > 5 and [r.seq for r in recs] == [r.seq for r in py]
    with pytest.raises(RuntimeError, match="does not fit"):
        _packed_records(str(plain), words_cap=512, reads_cap=16)                  # a 48 kb read cannot fit 8 kb of bases


@pytest.mark.skipif(seqio._load_io() is None, reason="libtopsicle_io.so not built")
def test_packed_reader_thread_team_resynchronises_at_any_byte(tmp_path):
    """The packed reader cuts the text into one stretch per thread; every thread but the first has to find the first
    record that starts in its stretch.  Reads of 0 .. 700 bases, quality lines that begin with '@' or '+', CRLF records,
    no newline at the end / blank lines at the end, many batch sizes (stretch boundaries land everywhere), a batch cut
    short by the buffer caps in the middle of a thread's stretch, and an irregular record that hands the rest to the
    streaming decoder.  Runs in a child process: the team size and threshold are read once per process."""
    import subprocess
    import sys
    import textwrap
    code = textwrap.dedent(r"""
        import os, sys
        import numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        from test_seqio_native import _packed_records
        from topsicle_amd import seqio
        rng = np.random.default_rng(5)
        def rec(i, n, crlf):
            s = "".join("ACGTNacgt"[x] for x in rng.integers(0, 9, n))
            q = "".join(chr(33 + int(x)) for x in rng.integers(0, 60, n))
            if n and i %% 3 == 0: q = "@" + q[1:]
            if n and i %% 5 == 0: q = "+" + q[1:]
            nl = "\r\n" if crlf else "\n"
            return f"@r{i} d{nl}{s}{nl}+{nl}{q}{nl}", (f"r{i}", s, q)
        parts, want = zip(*[rec(i, int(rng.integers(0, 700)), i %% 11 == 0) for i in range(900)])
        tmp = %r
        for name, text, exp in (("a.fastq", "".join(parts).rstrip("\n"), list(want)),
                                ("b.fastq", "".join(parts) + "\n\n \n", list(want)),
                                ("c.fastq", "".join(parts[:500]) + "@w\nAC\nGT\n+\nII\nII\n" + "".join(parts[500:]),
                                 list(want[:500]) + [("w", "ACGT", "IIII")] + list(want[500:]))):
            path = os.path.join(tmp, name)
            with open(path, "w", newline="") as h:
                h.write(text)
            for words_cap, reads_cap in ((1 << 20, 4096), (9000, 4096), (3000, 64), (1 << 20, 37), (1100, 4096)):
                recs, nb = _packed_records(path, words_cap=words_cap, reads_cap=reads_cap)
                assert [(r.id, r.seq, r.qual) for r in recs] == exp, (name, words_cap, reads_cap)
        print("ok")
    """) % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(tmp_path))
    for threads in ("2", "7", "16"):
        env = dict(os.environ, TOPSICLE_IO_DEBUG=f"threads={threads},pack_min_span=1")
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (threads, r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.skipif(seqio._load_io() is None, reason="libtopsicle_io.so not built")
def test_packed_reader_switches_to_ascii_decoder_on_odd_records(tmp_path):
    """Plain FASTQ whose records stop being 4-line records half way (wrapped sequence, '+name' lines, lower case, N):
    the packed mmap path hands over to the streaming decoder at that record; nothing is lost or duplicated."""
    fq = tmp_path / "odd.fastq"
    body = b""
    for i in range(40):
        body += b"@r%d\n" % i + b"ACGTNacgtn"[: 4 + i % 6] * 37 + b"\n+\n" + b"I" * (37 * (4 + i % 6)) + b"\n"
    body += b"@wrapped x y\nACGTAC\nGTTT\n+wrapped\nIIIIII\nIIII\n"
    for i in range(40, 60):
        body += b"@r%d\n" % i + b"GATTACA" * 11 + b"\n+r%d\n" % i + b"#" * 77 + b"\n"
    fq.write_bytes(body)
    recs, _ = _packed_records(str(fq))
    py = list(seqio.read_records(str(fq)))
    assert len(recs) == 61 and [(r.id, r.seq, r.qual) for r in recs] == [(r.id, r.seq, r.qual) for r in py]
    fa = tmp_path / "x.fa"
    fa.write_text(">a 1\nACGT\nAC\n>b\nNNNNTTTT\n")
    recs, _ = _packed_records(str(fa))
    assert [(r.id, r.seq, r.qual) for r in recs] == [("a", "ACGTAC", None), ("b", "NNNNTTTT", None)]


@pytest.mark.skipif(seqio._load_io() is None, reason="libtopsicle_io.so not built")
def test_packed_batch_writes_records_like_biopython(tmp_path, gold_dir):
    import io
    demo = os.path.join(gold_dir, "demo_col0.fastq.gz")
    plain = tmp_path / "demo.fastq"
    with gzip.open(demo, "rb") as g:
        plain.write_bytes(g.read())
    want = io.StringIO()
    py = list(seqio.read_records(demo))
    for r in py[3:9]:
        seqio.write_record(want, r, "fastq")
    for path in (str(plain), demo):
        pool = seqio.BufferPool(2, 1 << 20, 4096)
        got = io.BytesIO()
        for pb in seqio.read_batches_packed(path, pool):
            pb.release()
            pb.write_records(got, range(3, 9), "fastq")
        assert got.getvalue().decode() == want.getvalue()
    want = io.StringIO()
    seqio.write_record(want, py[0], "fasta")
    got = io.BytesIO()
    for pb in seqio.read_batches_packed(demo, seqio.BufferPool(2, 1 << 20, 4096)):
        pb.write_records(got, [0], "fasta")
        pb.release()
    assert got.getvalue().decode() == want.getvalue()
    # a real file handle takes the native writer (writev from the mapped input, neighbouring records merged); records whose
    # text is not already in SeqIO.write's layout ('+' line that repeats the name, CRLF) are re-assembled piece by piece
    odd = tmp_path / "odd.fastq"
    odd.write_bytes(b"@a first\nACGT\n+\nIIII\n@b\nACGTAC\n+b\nIIIIII\n@c x\r\nAC\r\n+\r\nII\r\n@d\nA\n+\n#\n@e\nGG\n+\n!!\n")
    for path, pick in ((str(plain), [0, 1, 2, 7, 8, 20, 43]), (str(plain), list(range(44))), (str(odd), [0, 1, 2, 3, 4]), (str(odd), [1, 3])):
        recs = list(seqio.read_records(path))
        want = io.StringIO()
        for i in pick:
            seqio.write_record(want, recs[i], "fastq")
        outp = tmp_path / "out.fastq"
        with open(outp, "wb") as h:
            h.write(b"")
            n = 0
            for pb in seqio.read_batches_packed(path, seqio.BufferPool(2, 1 << 20, 4096)):
                pb.release()
                pb.write_records(h, [i - n for i in pick if n <= i < n + pb.n], "fastq")
                n += pb.n
        assert outp.read_bytes().decode() == want.getvalue(), (path, pick)


def test_engine_pool_keeps_input_order_with_several_contexts(tmp_path):
    """Several contexts pull batches from one queue (dynamic balancing); results come back in file order."""
    from emu_engine import EmuEngine
    import topsicle_oracle as orc
    from topsicle_amd import batch, hiplib, synth
    bases, offsets, _ = synth.make_reads(90, 1500, "CCCTAA", seed=4, tract_min=200, tract_max=900)
    fq = tmp_path / "r.fastq"
    with open(fq, "wb") as h:
        for i in range(90):
            s = bytes(bases[offsets[i]:offsets[i + 1]])
            h.write(b"@read%d\n" % i + s + b"\n+\n" + b"I" * len(s) + b"\n")
    pats = orc.kmer_table("CCCTAA", 4)
    pool = batch.EnginePool([EmuEngine(), EmuEngine(), EmuEngine()], pats)
    prm = hiplib.make_params(min_len=0, min_count=-1)
    ids, bkps, nb = [], [], 0
    for pb, res, _s, _r, _w in pool.scan_file(str(fq), prm, max_bases=16384):
        ids += pb.ids
        bkps += res["bkp"].tolist()
        nb += 1
    assert nb >= 9 and ids == [f"read{i}" for i in range(90)]
    for i in (0, 17, 89):
        seq = bytes(bases[offsets[i]:offsets[i + 1]]).decode()
        cs, ce = orc.trc_counts(seq, pats)
        tail = "forward" if max(cs) > max(ce) else "reverse"
        _, counts = orc.window_count_matrix(seq, tail, pats, 100, 6, 100, 20000)
        assert bkps[i] == orc.binseg_l2_exact(counts.sum(axis=1))


def test_classic_mac_line_ends_read_like_text_mode(tmp_path):
    """A file whose lines end in a lone CR: the reference reads its input in Python's text mode, where that is a line end.  The
    native reader takes such a file (its first line ends in a lone CR) through the streaming decoder with the same three line
    ends; plain and gzip'ed, FASTQ and FASTA, a CR exactly at the decoder's buffer boundary included."""
    rng = np.random.default_rng(3)
    recs = []
    for i in range(3000):
        L = int(rng.integers(1, 3000))
        s = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), L))
        recs.append(b"@r%d text\r" % i + s + b"\r+\r" + bytes(rng.integers(33, 74, L, dtype=np.uint8)) + b"\r")
    data = b"".join(recs)
    (tmp_path / "cr.fastq").write_bytes(data)
    with gzip.open(tmp_path / "cr.fastq.gz", "wb", compresslevel=1) as h:
        h.write(data)
    (tmp_path / "cr.fasta").write_bytes(b">a one\rACGT\rAC\r>b\rGG\r")
    for name in ("cr.fastq", "cr.fastq.gz", "cr.fasta"):
        path = str(tmp_path / name)
        py = [(r.id, r.description, r.seq) for r in seqio.read_records(path)]
        assert len(py) == (2 if name.endswith("fasta") else 3000)
        assert [(r.id, r.description, r.seq) for r in all_records(path)] == py
        got = []
        for pb in seqio.read_batches_packed(path, seqio.BufferPool(2, 1 << 20, 4096)):
            got += [(pb.read_id(i), pb.head(i), bytes(pb.seq_bytes(i)).decode()) for i in range(pb.n)]
            pb.release()
        assert got == py


def test_differential_fuzz_against_the_python_parser():
    """tests/reader_fuzz.py: random FASTQ / FASTA files (CRLF, blank lines, wrapped sequences, lower case, N, '@' and '+' in
    quality lines, plain / gzip / two members), intact and damaged, through the packed mmap reader and the streaming ASCII
    reader: an intact file reads exactly as the pure-Python parser reads it, a damaged one as a prefix of the same records."""
    import reader_fuzz
    same, prefix = reader_fuzz.run(60, seed=4)
    assert same >= 60 and same + prefix == 120


@pytest.mark.skipif(seqio._load_io() is None, reason="libtopsicle_io.so not built")
def test_bgzf_window_that_ends_at_a_quality_line_stays_packed(tmp_path, monkeypatch):
    """a window of inflated text that ends exactly behind a record's last quality character (its line end is the first
    byte of the next window) used to be accepted as a whole record; the next window then began with the left-over newline, the
    packed decoder gave up and the REST of the file went through the one-thread streaming decoder (correct records, silent large
    slow-down).  Every BGZF block here ends at such a place and every group is one block: all batches must still be packed ones."""
    rng = np.random.default_rng(5)
    recs, text, cuts = [], b"", []
    for i in range(120):
        seq = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), int(rng.integers(800, 1200))))
        qual = bytes(rng.integers(33, 74, len(seq), dtype=np.uint8))
        text += b"@r%d\n" % i + seq + b"\n+\n" + qual
        if i % 4 == 3:
            cuts.append(len(text))                       # the block ends behind the quality line, in front of its "\n"
        text += b"\n"
        recs.append(("r%d" % i, seq.decode()))
    p = tmp_path / "cut.fastq.gz"
    _write_bgzf(str(p), text, splits=cuts)
    seqio.io_option("bgzf_group", 1)          # one block per group
    pool = seqio.BufferPool(3, 1024, 64)
    got, kinds = [], []
    for pb in seqio.read_batches_packed(str(p), pool):
        kinds.append("packed" if pb.spans is not None else "ascii")
        got += [(pb.read_id(i), pb.seq_bytes(i).decode()) for i in range(pb.n)]
        pb.release()
    assert got == recs
    assert set(kinds) == {"packed"} and len(kinds) > 5, kinds


@pytest.mark.skipif(seqio._load_io() is None, reason="libtopsicle_io.so not built")
@pytest.mark.parametrize("nl", [b"\n", b"\r\n"])
def test_wrapped_fasta_takes_the_thread_team_decoder(tmp_path, monkeypatch, nl):
    """Round 4: FASTA records whose sequence is wrapped (60 / 80 / any columns; blank lines in between) are joined line by line and
    packed by the thread team, plain, gzip'ed and bgzip'ed -- the same records, bases and 2-bit words as the same reads on one line
    each; a record with padded lines still goes to the streaming decoder."""
    seqio.io_option("pack_min_span", 2000)    # the team on small files
    seqio.io_option("pargz_min", 0)
    seqio.io_option("threads", 5)
    rng = np.random.default_rng(11)
    seqs = [bytes(rng.choice(np.frombuffer(b"ACGTacgtN", np.uint8), int(L))) for L in list(rng.integers(0, 4000, 150)) + [60, 61, 120, 1]]
    one, wrapped = b"", b""
    for i, s in enumerate(seqs):
        one += b">r%d d\n" % i + s + b"\n"
        w = [60, 80, 7, 1000][i % 4]
        lines = [s[j:j + w] for j in range(0, len(s), w)] or [b""]
        if i % 9 == 0 and len(lines) > 2:
            lines.insert(2, b"")
        wrapped += b">r%d d" % i + nl + nl.join(lines) + nl
    import gzip

    def packed(path):
        pool = seqio.BufferPool(3, 1 << 16, 4096)
        out, kinds = [], set()
        for pb in seqio.read_batches_packed(str(path), pool):
            kinds.add("packed" if pb.spans is not None else "ascii")
            for i in range(pb.n):
                w0, L = int(pb.desc["word_off"][i]), int(pb.desc["len"][i])
                out.append((pb.head(i), pb.seq_bytes(i), pb.qual_bytes(i), np.asarray(pb.seq2[w0:w0 + (L + 15) // 16]).tobytes(),
                            int(pb.desc["flags"][i])))
            pb.release()
        return out, kinds
    (tmp_path / "one.fasta").write_bytes(one)
    want, kinds = packed(tmp_path / "one.fasta")
    assert kinds == {"packed"} and [x[1] for x in want] == seqs and all(x[2] is None for x in want)
    (tmp_path / "w.fasta").write_bytes(wrapped)
    (tmp_path / "w.fasta.gz").write_bytes(gzip.compress(wrapped, 4))
    _write_bgzf(str(tmp_path / "wb.fasta.gz"), wrapped, block=3001)
    for name in ("w.fasta", "w.fasta.gz", "wb.fasta.gz"):
        got, kinds = packed(tmp_path / name)
        assert kinds == {"packed"}, (name, kinds)
        assert got == want, name
    (tmp_path / "pad.fasta").write_bytes(b">a\nACGT \n  AC\n>b\nTTTT\n")
    got, kinds = packed(tmp_path / "pad.fasta")
    assert [x[1] for x in got] == [b"ACGTAC", b"TTTT"] and kinds == {"ascii"}


@pytest.mark.skipif(seqio._load_io() is None, reason="libtopsicle_io.so not built")
@pytest.mark.parametrize("nl", [b"\n", b"\r\n"])
def test_multiline_fastq_takes_the_thread_team_decoder(tmp_path, monkeypatch, nl):
    """Round 4: FASTQ records whose sequence and quality are spread over several lines -- each at its own
    width, quality lines that begin with '@' or '+', a name repeated on the '+' line -- are joined and packed by the thread team,
    plain, gzip'ed and bgzip'ed: the same records, qualities and 2-bit words as the same reads written four lines each; the
    record writer puts them out in Biopython's four-line layout; a record with a blank line inside still goes to the streaming
    decoder."""
    seqio.io_option("pack_min_span", 2000)    # the team on small files
    seqio.io_option("pargz_min", 0)
    seqio.io_option("threads", 5)
    rng = np.random.default_rng(12)
    seqs = [bytes(rng.choice(np.frombuffer(b"ACGTacgtN", np.uint8), int(L))) for L in list(rng.integers(1, 4000, 150)) + [60, 61, 120, 1, 2]]
    quals = [bytes(rng.choice(np.frombuffer(b"@+I#5", np.uint8), len(s))) for s in seqs]
    four, multi = b"", b""
    for i, (s, q) in enumerate(zip(seqs, quals)):
        plus = b"+r%d d" % i if i % 7 == 0 else b"+"
        four += b"@r%d d\n" % i + s + b"\n+\n" + q + b"\n"
        ws, wq = [60, 80, 7, 1000][i % 4], [60, 61, 13, 999][i % 4]
        multi += (b"@r%d d" % i + nl + nl.join(s[j:j + ws] for j in range(0, len(s), ws)) + nl + plus + nl +
                  nl.join(q[j:j + wq] for j in range(0, len(q), wq)) + nl)
    import gzip

    def packed(path, write_to=None):
        pool = seqio.BufferPool(3, 1 << 16, 4096)
        out, kinds = [], set()
        for pb in seqio.read_batches_packed(str(path), pool):
            kinds.add("packed" if pb.spans is not None else "ascii")
            for i in range(pb.n):
                w0, L = int(pb.desc["word_off"][i]), int(pb.desc["len"][i])
                out.append((pb.head(i), bytes(pb.seq_bytes(i)), bytes(pb.qual_bytes(i)), np.asarray(pb.seq2[w0:w0 + (L + 15) // 16]).tobytes(),
                            int(pb.desc["flags"][i])))
            if write_to is not None:
                pb.write_records(write_to, list(range(0, pb.n, 2)), "fastq")
            pb.release()
        return out, kinds
    (tmp_path / "four.fastq").write_bytes(four)
    want, kinds = packed(tmp_path / "four.fastq")
    assert kinds == {"packed"} and [x[1] for x in want] == seqs and [x[2] for x in want] == quals
    (tmp_path / "m.fastq").write_bytes(multi)
    (tmp_path / "m.fastq.gz").write_bytes(gzip.compress(multi, 4))
    _write_bgzf(str(tmp_path / "mb.fastq.gz"), multi, block=3001)
    for name in ("m.fastq", "m.fastq.gz", "mb.fastq.gz"):
        with open(tmp_path / ("out_" + name + ".fq"), "wb") as h:
            got, kinds = packed(tmp_path / name, write_to=h)
        assert kinds == {"packed"}, (name, kinds)
        assert got == want, name
        written = list(seqio.read_records(str(tmp_path / ("out_" + name + ".fq"))))
        assert len(written) >= len(seqs) // 2 and all(b"\n" not in r.seq.encode() for r in written)
        by_id = {"r%d" % i: (s.decode(), q.decode()) for i, (s, q) in enumerate(zip(seqs, quals))}
        assert all((r.seq, r.qual) == by_id[r.id] for r in written)
        assert (tmp_path / ("out_" + name + ".fq")).read_bytes().count(b"\n") == 4 * len(written)
    (tmp_path / "blank.fastq").write_bytes(b"@a\nACGT\n\nAC\n+\nIIII\nII\n@b\nTTTT\n+\n####\n")
    got, kinds = packed(tmp_path / "blank.fastq")
    py = list(seqio.read_records(str(tmp_path / "blank.fastq")))
    assert [(x[1].decode(), x[2].decode()) for x in got] == [(r.seq, r.qual) for r in py] and kinds == {"ascii"}


@pytest.mark.skipif(seqio._load_io() is None, reason="libtopsicle_io.so not built")
def test_compressed_windows_are_inflated_and_indexed_only_for_who_uses_them(tmp_path):
    """an ASCII consumer of bgzip'ed FASTA -- which the thread-team decoder does not serve -- used to inflate and
    line-index a whole first window at open and throw it away; the packed decoder line-indexed every window it never looked at by
    lines.  Now the first window is built by the first call that wants records, and the line index only for Fast::next.  The
    library's own timing lines ($TOPSICLE_IO_DEBUG=timing in a fresh process) say what ran."""
    import subprocess
    import sys
    rng = np.random.default_rng(5)
    fa = b"".join(b">r%d\n" % i + bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), 3000)) + b"\n" for i in range(300))
    fq = b"".join(b"@r%d\n" % i + bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), 3000)) + b"\n+\n" + b"I" * 3000 + b"\n" for i in range(300))
    _write_bgzf(str(tmp_path / "a.fasta.gz"), fa)
    _write_bgzf(str(tmp_path / "q.fastq.gz"), fq)
    code = """
import sys
sys.path.insert(0, %r)
from topsicle_amd import seqio
mode, path = sys.argv[1], sys.argv[2]
n = 0
if mode == "ascii":
    for b in seqio.read_batches(path):
        n += len(b)
else:
    pool = seqio.BufferPool(2, 1 << 18, 4096)
    for pb in seqio.read_batches_packed(path, pool):
        assert pb.spans is not None
        n += pb.n
        pb.release()
print("records", n)
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(mode, name):
        r = subprocess.run([sys.executable, "-c", code, mode, str(tmp_path / name)], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, TOPSICLE_IO_DEBUG="timing"))
        assert r.returncode == 0 and "records 300" in r.stdout, r.stdout + r.stderr
        return r.stderr
    err = run("ascii", "a.fasta.gz")
    assert "bgzf group" not in err and "[tps_io] index" not in err, err       # nothing inflated by the team, nothing indexed
    err = run("packed", "a.fasta.gz")
    assert "bgzf group" in err and "[tps_io] index" not in err, err
    err = run("packed", "q.fastq.gz")
    assert "bgzf group" in err and "[tps_io] index" not in err, err
    err = run("ascii", "q.fastq.gz")
    assert "bgzf group" in err and "[tps_io] index" in err, err               # Fast::next works on the line index
