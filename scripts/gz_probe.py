"""Diagnostic: how the parallel gzip inflater (csrc/tps_gzpar.h) scales with threads on this host, on a level-1 gzip of a
synthetic ONT-like FASTQ file, next to one-stream zlib.  python scripts/gz_probe.py [n_reads] [threads ...]"""
import ctypes as C
import os
import subprocess
import sys
import tempfile
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from topsicle_amd import seqio, synth  # noqa: E402


def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    threads = [int(x) for x in sys.argv[2:]] or [1, 4, 8, 16, 32, 64]
    lib = seqio._load_io()
    lib.tps_gz_inflate.restype = C.c_int64
    lib.tps_gz_inflate.argtypes = [C.c_char_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_void_p]
    tmp = tempfile.mkdtemp(prefix="gzprobe_")
    fq = os.path.join(tmp, "r.fastq")
    bases, offsets, _ = synth.make_reads(n_reads, 15000, "CCCTAA", seed=3, errors=synth.ONT)
    rng = np.random.default_rng(1)
    with open(fq, "wb") as h:
        for i in range(n_reads):
            s = bases[offsets[i]:offsets[i + 1]].tobytes()
            h.write(b"@r%d\n" % i + s + b"\n+\n" + (rng.integers(0, 30, len(s), dtype=np.uint8) + 40).tobytes() + b"\n")
    subprocess.check_call(["gzip", "-1", "-k", "-f", fq])
    gz = fq + ".gz"
    print("text", os.path.getsize(fq), "gz", os.path.getsize(gz), "cpus", len(os.sched_getaffinity(0)), flush=True)
    t = time.perf_counter()
    d, tot = zlib.decompressobj(31), 0
    with open(gz, "rb") as h:
        while True:
            b = h.read(1 << 22)
            if not b:
                break
            tot += len(d.decompress(b))
    t_z = time.perf_counter() - t
    print("zlib one stream: %.3f s  %.0f MB/s" % (t_z, tot / t_z / 1e6), flush=True)
    stats = np.zeros(3, np.int64)
    for th in threads:
        best = 1e9
        for _ in range(3):
            t = time.perf_counter()
            n = lib.tps_gz_inflate(gz.encode(), None, 0, th, 0, stats.ctypes.data)
            best = min(best, time.perf_counter() - t)
        print("threads %2d: %.3f s  %.0f MB/s text  x%.1f of zlib  chunks/spec/serial %s" % (th, best, n / best / 1e6, t_z / best, stats), flush=True)
    for f in (fq, gz):
        os.unlink(f)
    os.rmdir(tmp)


if __name__ == "__main__":
    main()
