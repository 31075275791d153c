"""The parallel gzip inflater of libtopsicle_io.so (csrc/tps_gzpar.h: speculative block starts, 16-bit symbols with window
markers, in-order stitching, CRC check) against zlib on the same files: FASTQ-like text at several levels, every deflate
block type (stored, fixed, dynamic), strategies that produce odd Huffman trees, long-distance and overlapping matches,
several members, many small rounds; corrupt and truncated files are errors, never wrong text."""
import ctypes as C
import gzip
import os
import zlib

import numpy as np
import pytest

from topsicle_amd import seqio


def _lib():
    lib = seqio._load_io()
    lib.tps_gz_inflate.restype = C.c_int64
    lib.tps_gz_inflate.argtypes = [C.c_char_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_void_p]
    return lib


def inflate(path, threads=8, want=0, cap=None):
    lib = _lib()
    cap = cap if cap is not None else 1 << 28
    out = np.empty(cap, np.uint8)
    stats = np.zeros(3, np.int64)
    n = lib.tps_gz_inflate(str(path).encode(), out.ctypes.data, cap, threads, want, stats.ctypes.data)
    return (None if n < 0 else out[:n].tobytes()), stats


def gz_bytes(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, wbits=31, memlevel=8):
    co = zlib.compressobj(level, zlib.DEFLATED, wbits, memlevel, strategy)
    return co.compress(data) + co.flush()


def fastq_text(n_reads, read_len, seed):
    rng = np.random.default_rng(seed)
    parts = []
    for i in range(n_reads):
        L = int(read_len * (0.5 + rng.random()))
        s = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, L)]
        if rng.random() < 0.5:                                   # telomere-like repeats: long matches
            s[: L // 3] = np.frombuffer((b"CCCTAA" * (L // 18 + 1))[: L // 3], np.uint8)
        q = (33 + rng.integers(0, 40, L)).astype(np.uint8)
        parts.append(b"@read%d len=%d\n" % (i, L) + s.tobytes() + b"\n+\n" + q.tobytes() + b"\n")
    return b"".join(parts)


@pytest.fixture(scope="module")
def text():
    return fastq_text(2500, 12000, 7)                            # ~60 MB


@pytest.mark.parametrize("level", [1, 6, 9])
def test_fastq_text_levels(tmp_path, text, level):
    p = tmp_path / f"l{level}.fastq.gz"
    p.write_bytes(gz_bytes(text, level))
    got, stats = inflate(p, threads=8, want=8 << 20)
    assert got == text
    assert stats[0] > 8 and stats[1] >= 0.6 * (stats[0] - stats[0] // 8)      # most speculative chunks were found and fitted
    got1, _ = inflate(p, threads=1)
    assert got1 == text


def test_block_types_and_strategies(tmp_path, text):
    rng = np.random.default_rng(1)
    small = text[: 6 << 20]
    cases = {
        "stored": gz_bytes(rng.integers(0, 256, 5 << 20, dtype=np.uint8).tobytes(), 6),          # incompressible: stored blocks
        "level0": gz_bytes(small, 0),
        "fixed": gz_bytes(small, 6, zlib.Z_FIXED),
        "huffman_only": gz_bytes(small, 6, zlib.Z_HUFFMAN_ONLY),
        "rle": gz_bytes(small, 6, zlib.Z_RLE),
        "filtered": gz_bytes(small, 9, zlib.Z_FILTERED),
        "memlevel1": gz_bytes(small, 6, memlevel=1),                                             # many tiny blocks
        "runs": gz_bytes(b"A" * (9 << 20) + b"CG" * (2 << 20) + small[: 1 << 20], 9),            # distance-1 / distance-2 overlapping matches
        "mixed": gz_bytes(small[: 2 << 20] + rng.integers(0, 256, 2 << 20, dtype=np.uint8).tobytes() + small[2 << 20: 4 << 20], 6),
    }
    for name, blob in cases.items():
        p = tmp_path / (name + ".gz")
        p.write_bytes(blob)
        want = gzip.decompress(blob)
        for threads, w in ((8, 1 << 20), (3, 4 << 20), (1, 0)):
            got, stats = inflate(p, threads=threads, want=w)
            assert got == want, (name, threads, w, stats)


def test_members_padding_and_small_files(tmp_path, text):
    a, b, c = text[: 3 << 20], text[3 << 20: 3 << 20 | 5], text[5 << 20: 9 << 20]
    blob = gz_bytes(a, 6) + gz_bytes(b, 9) + gz_bytes(b"", 6) + gz_bytes(c, 1)
    p = tmp_path / "multi.gz"
    p.write_bytes(blob)
    assert inflate(p, threads=4, want=1 << 20)[0] == a + b + c
    p.write_bytes(blob + b"\0" * 100)                            # zero padding behind the last member is ignored, like gzip does
    assert inflate(p, threads=4)[0] == a + b + c
    # header with name + extra field
    import io
    bio = io.BytesIO()
    with gzip.GzipFile(filename="reads.fastq", fileobj=bio, mode="wb", compresslevel=6) as g:
        g.write(a)
    p.write_bytes(bio.getvalue())
    assert inflate(p, threads=4, want=1 << 20)[0] == a
    for data in (b"", b"x", b"@r\nACGT\n+\nIIII\n"):
        p.write_bytes(gz_bytes(data))
        assert inflate(p)[0] == data


def test_corrupt_and_truncated_files_are_errors(tmp_path, text):
    blob = bytearray(gz_bytes(text[: 8 << 20], 6))
    p = tmp_path / "bad.gz"
    for cut in (len(blob) - 3, len(blob) // 2, 40):
        p.write_bytes(bytes(blob[:cut]))
        assert inflate(p, threads=4, want=1 << 20)[0] is None
    rng = np.random.default_rng(5)
    bad = 0
    for _ in range(6):
        b2 = bytearray(blob)
        pos = int(rng.integers(100, len(b2) - 100))
        b2[pos] ^= 1 << int(rng.integers(8))
        p.write_bytes(bytes(b2))
        got, _ = inflate(p, threads=4, want=1 << 20)
        assert got is None                                       # a flipped bit never yields text (CRC)
        bad += 1
    assert bad == 6


def test_reader_takes_the_parallel_path_for_fastq_gz(tmp_path, text):
    """seqio on an ordinary .fastq.gz: same records as on the plain file (the reader inflates with the thread team)."""
    plain = tmp_path / "r.fastq"
    plain.write_bytes(text)
    gz = tmp_path / "r.fastq.gz"
    gz.write_bytes(gz_bytes(text, 6))
    a = [(r.id, r.seq, r.qual) for rb in seqio.read_batches(str(plain), max_bases=8 << 20) for r in rb.records()] if hasattr(seqio.RecordBatch, "records") else None
    ids_plain = [i for pb in seqio.read_batches_packed(str(plain), seqio.BufferPool(2, 1 << 22, 1 << 16)) for i in pb.ids]
    ids_gz, nb = [], 0
    for pb in seqio.read_batches_packed(str(gz), seqio.BufferPool(2, 1 << 22, 1 << 16)):
        ids_gz += pb.ids
        nb += pb.n_bases
    assert ids_gz == ids_plain and len(ids_gz) == 2500
    seqio.io_option("no_pargz", 1)
    try:
        ids_z = [i for pb in seqio.read_batches_packed(str(gz), seqio.BufferPool(2, 1 << 22, 1 << 16)) for i in pb.ids]
    finally:
        seqio.io_option("no_pargz", 0)
    assert ids_z == ids_plain


def test_clmul_crc32_equals_zlib():
    """The carry-less-multiplication CRC-32 the inflater checks members with (tps_gzpar.h: crc32_fast) against zlib.crc32:
    every length around the 16- and 64-byte folding steps, unaligned starts, chained calls, a long buffer."""
    lib = seqio._load_io()
    lib.tps_crc32.restype = C.c_uint32
    lib.tps_crc32.argtypes = [C.c_uint32, C.c_void_p, C.c_int64]
    rng = np.random.default_rng(7)
    buf = rng.integers(0, 256, (3 << 20) + 77, dtype=np.uint8)
    raw = buf.tobytes()
    for n in list(range(0, 700)) + [1023, 1024, 1025, 4096 + 15, 65536 + 48, len(raw)]:
        for off in (0, 1, 3, 13):
            if off + n > len(raw):
                continue
            assert lib.tps_crc32(0, buf.ctypes.data + off, n) == zlib.crc32(raw[off:off + n]), (n, off)
    c1, c2, pos = 0, 0, 0
    for n in (5, 300, 64, 1000, 16, 257, 100000, 3):     # chained: the running value goes in and out
        c1 = lib.tps_crc32(c1, buf.ctypes.data + pos, n)
        c2 = zlib.crc32(raw[pos:pos + n], c2)
        pos += n
        assert c1 == c2


@pytest.mark.parametrize("level", [1, 6, 9])
def test_short_period_runs(tmp_path, level):
    """Matches whose distance is below one 8-byte word (periods 1 .. 7 bytes, 1 .. 3 symbols in a speculative chunk): the fast
    symbol loop copies them as words at a multiple of the period."""
    rng = np.random.default_rng(level)
    parts = []
    for rep in range(3000):
        period = int(rng.integers(1, 12))
        unit = bytes(rng.integers(65, 70, period, dtype=np.uint8))
        parts.append(unit * int(rng.integers(1, 3000 // period + 2)))
        parts.append(bytes(rng.integers(33, 127, int(rng.integers(0, 1200)), dtype=np.uint8)))   # (noise: the compressed file is large enough to be cut)
    data = b"".join(parts)
    p = tmp_path / "runs.gz"
    p.write_bytes(gz_bytes(data, level))
    spec = 0
    for threads, want in ((1, 0), (4, 1 << 16), (8, 1 << 18)):
        text, stats = inflate(p, threads=threads, want=want)
        assert text == data
        spec += int(stats[1])
    assert spec > 0                                      # speculative chunks (16-bit symbols) took part


def test_differential_fuzz_against_zlib():
    """tests/gz_fuzz.py: random texts, random deflate parameters and flushes, random damage (bit flips, truncation, trailing
    bytes) -- the text zlib gives or an error, never anything else.  Seed 1's case 11 is the damaged stream whose speculative
    chunk decoded garbage without ever reaching a block end and asked for 78 GB (growth is bounded where the buffer grows now)."""
    import gz_fuzz
    ok, err = gz_fuzz.run(16, seed=1)
    assert ok >= 1 and err >= 3 and ok + err >= 8
