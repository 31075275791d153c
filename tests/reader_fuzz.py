"""Differential fuzz of the native FASTQ / FASTA reader + packer (csrc/tps_io.cpp) against the pure-Python parser of
topsicle_amd/seqio.py -- run by hand, best on the sanitizer build (see tests/gz_fuzz.py for the environment):
    ... TOPSICLE_IO_LIB=tests/emu/_build/libtopsicle_io_asan.so python tests/reader_fuzz.py [cases [seed]]
Random well-formed files (CRLF, blank lines, wrapped sequences, lower case, N, '@' and '+' inside quality lines, names on the
'+' line, plain / gzip / several members) and damaged ones (lines dropped, cut short, bytes changed): whatever the Python
parser reads before it stops, the native readers -- the packed mmap path and the streaming ASCII path -- must read the same
records in the same order (on a damaged file: a prefix of the same records, or the same records); never a crash."""
import gzip
import logging
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topsicle_amd import seqio  # noqa: E402


def make_file(rng, path):
    fastq = rng.random() < 0.75
    u = rng.random()
    nl = b"\r\n" if u < 0.15 else b"\r" if u < 0.22 else b"\n"          # (CR alone: classic Mac OS text, a line end in Python's text mode)
    n = int(rng.integers(0, 60))
    out = []
    for i in range(n):
        L = int(rng.choice([0, 1, 5, 60, 61, 200, 1000, 5000, 70000])) if rng.random() < 0.5 else int(rng.integers(0, 3000))
        alphabet = b"ACGT" if rng.random() < 0.7 else b"ACGTNacgtn"
        seq = bytes(rng.choice(np.frombuffer(alphabet, np.uint8), L)) if L else b""
        name = b"r%d" % i + (b" some text" if rng.random() < 0.5 else b"")
        if fastq:
            qual = bytes(rng.integers(33, 75, L, dtype=np.uint8)) if L else b""      # ('@' = 64 and '+' = 43 occur, also first)
            plus = b"+" + (name if rng.random() < 0.1 else b"")
            if L and rng.random() < 0.3:           # (round 4) a multi-line record: sequence and quality wrapped at their own widths
                ws, wq = (int(rng.choice([60, 80, int(rng.integers(1, 150))])) for _ in range(2))
                out.append(b"@" + name + nl + nl.join(seq[j:j + ws] for j in range(0, L, ws)) + nl + plus + nl +
                           nl.join(qual[j:j + wq] for j in range(0, L, wq)) + nl)
            else:
                out.append(b"@" + name + nl + seq + nl + plus + nl + qual + nl)
        else:
            w = int(rng.choice([60, 80, 10 ** 9, int(rng.integers(1, 150))]))       # (round 4: any line width -- wrapped FASTA takes the thread-team decoder)
            lines = [seq[j:j + w] for j in range(0, len(seq), w)] or [b""]
            if rng.random() < 0.03 and len(lines) > 1:
                lines.insert(int(rng.integers(1, len(lines))), b"")                 # a blank line inside a record
            out.append(b">" + name + nl + nl.join(lines) + nl)
        if rng.random() < 0.05:
            out.append(nl)
    data = b"".join(out)
    damage = int(rng.integers(5))
    if damage == 1 and len(data) > 10:
        data = data[:int(rng.integers(1, len(data)))]
    elif damage == 2 and len(data) > 10:
        lines = data.split(nl)
        del lines[int(rng.integers(len(lines)))]
        data = nl.join(lines)
    elif damage == 3 and len(data) > 10:
        b = bytearray(data)
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(len(b)))] = int(rng.choice(np.frombuffer(b"@+>A " + (b"\r" if nl == b"\r" else b"\n"), np.uint8)))   # (no lone CR in LF text, no LF in CR text: in a mixed file a lone CR is a line end for text mode and data for the native reader)
        data = bytes(b)
    ext = ".fastq" if fastq else ".fasta"
    mode = int(rng.integers(3))
    if mode == 0:
        p = path + ext
        with open(p, "wb") as h:
            h.write(data)
    elif mode == 1:
        p = path + ext + ".gz"
        with gzip.open(p, "wb", compresslevel=int(rng.integers(1, 10))) as h:
            h.write(data)
    elif mode == 2 and rng.random() < 0.5:
        p = path + ext + ".gz"
        with open(p, "wb") as h:                                  # several members
            cut = int(rng.integers(0, len(data) + 1))
            h.write(gzip.compress(data[:cut]) + gzip.compress(data[cut:]))
    else:
        p = path + ext + ".gz"                                    # BGZF: independent blocks with the 'BC' extra field
        import struct
        import zlib

        def member(chunk):
            co = zlib.compressobj(int(rng.integers(1, 10)), zlib.DEFLATED, -15)
            body = co.compress(chunk) + co.flush()
            bsize = 12 + 6 + len(body) + 8
            return (b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1) +
                    body + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
        block = int(rng.choice([300, 5000, 60000]))
        with open(p, "wb") as h:
            for lo in range(0, len(data), block):
                h.write(member(data[lo:lo + block]))
            h.write(member(b""))
    return p, damage


def run(cases=300, seed=0):
    logging.disable(logging.CRITICAL)
    tmp = tempfile.mkdtemp(prefix="rdfuzz_")
    same = prefix = 0
    for case in range(cases):
        rng = np.random.default_rng(seed * 1000003 + case)
        p, damage = make_file(rng, os.path.join(tmp, "f%d" % case))
        # compressed inputs: the thread team's inflaters also on small files, windows of inflated text from tiny to default
        seqio.io_option("pargz_min", int(rng.choice([0, 1 << 40])))
        g = int(rng.choice([0, 700, 20000, 1 << 20]))
        if g:
            seqio.io_option("bgzf_group", g)
        else:
            seqio.io_option("bgzf_group", 0)
        if os.environ.get("RD_FUZZ_VERBOSE"):
            print("case", case, p, flush=True)
        want = [(r.id, r.description, r.seq, r.qual) for r in seqio.read_records(p)]
        got_a = []
        for b in seqio.read_batches(p, max_bases=int(rng.choice([1 << 12, 1 << 16, 1 << 24])), max_records=int(rng.choice([3, 64, 1 << 20]))):
            got_a += [(r.id, r.description, r.seq, r.qual) for r in (b.record(i) for i in range(len(b)))]
        got_p = []
        pool = seqio.BufferPool(2, int(rng.choice([1 << 14, 1 << 18])), int(rng.choice([8, 4096])))
        try:
            for pb in seqio.read_batches_packed(p, pool):
                for i in range(pb.n):
                    sq = bytes(pb.seq_bytes(i))
                    ql = pb.qual_bytes(i)
                    got_p.append((pb.read_id(i), pb.head(i), sq.decode("latin1"), None if ql is None else bytes(ql).decode("latin1")))
                    # what was PACKED is that sequence (2-bit codes (c >> 1) & 3, invalid mask = not one of acgtACGT)
                    w0, L = int(pb.desc["word_off"][i]), int(pb.desc["len"][i])
                    assert L == len(sq), (case, p, i, L, len(sq))
                    nw = (L + 15) // 16
                    words = np.asarray(pb.seq2[w0:w0 + nw], np.uint32)
                    codes = ((words[:, None] >> (2 * np.arange(16, dtype=np.uint32))) & 3).reshape(-1)[:L]
                    raw = np.frombuffer(sq, np.uint8)
                    assert np.array_equal(codes, (raw >> 1) & 3), (case, p, i, "packed bases differ from the record's sequence")
                    if pb.inv is not None:
                        iv = np.asarray(pb.inv[w0:w0 + nw], np.uint16)
                        bad = ((iv[:, None] >> np.arange(16, dtype=np.uint16)) & 1).reshape(-1)[:L].astype(bool)
                        assert np.array_equal(bad, ~np.isin(raw, np.frombuffer(b"ACGTacgt", np.uint8))), (case, p, i, "invalid mask")
                pb.release()
        except RuntimeError as e:                                 # a read larger than the (deliberately small) upload buffers
            if "does not fit" not in str(e):
                raise
            got_p = None
        for name, got in (("ascii", got_a), ("packed", got_p)):
            if got is None:
                continue
            g = [(a, b, c.upper(), d) for a, b, c, d in got] if name == "packed" else got
            w = [(a, b, c.upper(), d) for a, b, c, d in want] if name == "packed" else want
            if g == w:
                same += 1
            else:
                # a damaged file: the readers may stop at different records, but what they deliver is the same stream
                k = min(len(g), len(w))
                assert damage != 0, (case, name, p, "an intact file read differently", len(g), len(w))
                assert g[:k] == w[:k], (case, name, p, len(g), len(w))
                prefix += 1
        os.unlink(p)
    os.rmdir(tmp)
    seqio.io_option("pargz_min", -1)
    seqio.io_option("bgzf_group", 0)
    return same, prefix


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    same, prefix = run(cases, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print(f"{cases} files: {same} reader passes identical to the Python parser, {prefix} stopped at a different record of a damaged file")
