"""Differential fuzz of the parallel gzip inflater (csrc/tps_gzpar.h) against zlib -- run by hand, best on the sanitizer build:
    LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 PYTHONMALLOC=malloc \\
    TOPSICLE_IO_LIB=tests/emu/_build/libtopsicle_io_asan.so python tests/gz_fuzz.py [cases [seed]]
Random texts (noise, runs, FASTQ-like, periodic), random deflate parameters, then random damage (bit flips, truncation, garbage
appended): the inflater must return exactly what zlib returns, or an error when zlib fails -- never other text, never a crash."""
import ctypes as C
import os
import sys
import tempfile
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topsicle_amd import seqio  # noqa: E402


def run(cases=300, seed=0):
    lib = seqio._load_io()
    lib.tps_gz_inflate.restype = C.c_int64
    lib.tps_gz_inflate.argtypes = [C.c_char_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_void_p]
    tmp = tempfile.mkdtemp(prefix="gzfuzz_")
    path = os.path.join(tmp, "f.gz")
    out = np.empty(64 << 20, np.uint8)
    stats = np.zeros(3, np.int64)
    n_ok = n_err = 0
    for case in range(cases):
        rng = np.random.default_rng(seed * 1000003 + case)
        if os.environ.get('GZ_FUZZ_VERBOSE'): print('case', case, flush=True)
        parts = []
        for _ in range(int(rng.integers(1, 40))):
            kind = int(rng.integers(5))
            n = int(rng.integers(1, 300000))
            if kind == 0:
                parts.append(bytes(rng.integers(0, 256, n, dtype=np.uint8)))
            elif kind == 1:
                parts.append(bytes(rng.integers(65, 69, int(rng.integers(1, 9)), dtype=np.uint8)) * (n // 4 + 1))
            elif kind == 2:
                parts.append(bytes(rng.choice(np.frombuffer(b"ACGT\n", np.uint8), n)))
            elif kind == 3:
                parts.append(bytes(rng.integers(33, 74, n, dtype=np.uint8)))
            else:
                parts.append(bytes([int(rng.integers(256))]) * n)
        data = b"".join(parts)
        co = zlib.compressobj(int(rng.integers(0, 10)), zlib.DEFLATED, 31, int(rng.integers(1, 10)),
                              int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED])))
        gz = bytearray()
        pos = 0
        while pos < len(data):                                  # (flushes: stored / empty blocks in between, several members at times)
            step = int(rng.integers(1, len(data) + 1))
            gz += co.compress(data[pos:pos + step])
            if rng.random() < 0.3:
                gz += co.flush(int(rng.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH])))
            pos += step
        gz += co.flush()
        damage = int(rng.integers(4))
        if damage == 1 and len(gz) > 30:
            for _ in range(int(rng.integers(1, 4))):
                i = int(rng.integers(10, len(gz)))
                gz[i] ^= 1 << int(rng.integers(8))
        elif damage == 2 and len(gz) > 30:
            del gz[int(rng.integers(12, len(gz))):]
        elif damage == 3:
            gz += bytes(rng.integers(0, 256, int(rng.integers(1, 64)), dtype=np.uint8))
        with open(path, "wb") as h:
            h.write(gz)
        try:
            d = zlib.decompressobj(31)
            want = d.decompress(bytes(gz))
            if not d.eof:
                want = None                                     # truncated
            elif d.unused_data.strip(b"\0"):
                want = "garbage"                                # something behind the member: another member or garbage (gzip tools differ)
        except zlib.error:
            want = None
        for threads, w in ((int(rng.integers(1, 9)), int(rng.choice([0, 1 << 16, 1 << 20]))),):
            n = lib.tps_gz_inflate(path.encode(), out.ctypes.data, len(out), threads, w, stats.ctypes.data)
            if want is None:
                assert n < 0, (case, "accepted a stream zlib rejects", n)
                n_err += 1
            elif want == "garbage":
                assert n < 0 or out[:n].tobytes()[:len(data)] == data[:n], (case, "wrong text in front of trailing bytes")
            else:
                assert n == len(want) and out[:n].tobytes() == want, (case, "text differs", n, len(want))
                n_ok += 1
    os.unlink(path)
    os.rmdir(tmp)
    return n_ok, n_err


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    ok, err = run(cases, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print(f"{cases} cases: {ok} equal to zlib, {err} rejected like zlib")
