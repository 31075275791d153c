"""Reader alone (no GPU work): packed batches of a config-2-shaped plain FASTQ file, best of a few passes.
TOPSICLE_IO_LIB selects the library, TOPSICLE_IO_DEBUG=timing prints the reader's phase times.  usage: reader_bench.py [n_reads]"""
import os, sys, time
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT)
from topsicle_amd import synth, e2e, seqio, batch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
b, o, _ = synth.make_reads(n, 15000, "CCCTAA", seed=1)
d = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
fq = os.path.join(d, "tps_reader_bench_%d.fastq" % os.getpid())
e2e.write_fastq(fq, b, o)
try:
    pool = seqio.BufferPool(4, batch.BATCH_BASES // 16, min(batch.BATCH_READS, batch.BATCH_BASES // 64), None)
    best = None
    for rep in range(5):
        t0 = time.perf_counter(); nb = 0
        for pb in seqio.read_batches_packed(fq, pool):
            nb += pb.n_bases; pb.release()
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
        print("pass %d: %.2f ms, %.3e bases/s" % (rep, dt * 1e3, nb / dt), flush=True)
    print("best %.2f ms = %.3e bases/s (%s)" % (best * 1e3, nb / best, os.environ.get("TOPSICLE_IO_LIB", "libtopsicle_io.so")))
finally:
    os.remove(fq)
