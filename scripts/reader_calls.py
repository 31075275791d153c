"""Diagnostic: wall time of the reader's individual native calls on a config-2-shaped plain FASTQ (pinned buffers if a GPU is there)."""
import ctypes as C, os, sys, time
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT)
import numpy as np
from topsicle_amd import synth, e2e, seqio, batch
b, o, _ = synth.make_reads(10000, 15000, "CCCTAA", seed=1)
d = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
fq = os.path.join(d, "tps_reader_calls_%d.fastq" % os.getpid())
e2e.write_fastq(fq, b, o)
alloc = None
try:
    from topsicle_amd import hiplib
    eng = hiplib.HipScanner(0)
    alloc = eng.host_alloc
except Exception as e:
    print("no GPU context:", e)
lib = seqio._load_io()
pool = seqio.BufferPool(4, batch.BATCH_BASES // 16, min(batch.BATCH_READS, batch.BATCH_BASES // 64), alloc)
try:
    for rep in range(4):
        T = []
        t00 = time.perf_counter()
        t0 = time.perf_counter(); h = C.c_void_p(); lib.tps_reader_open(fq.encode(), C.byref(h)); T.append(("open", time.perf_counter() - t0))
        while True:
            bs = pool.get(); nw = C.c_int64(0)
            t1 = time.perf_counter()
            n = lib.tps_reader_next_packed(h, bs.seq2.ctypes.data, bs.inv.ctypes.data, bs.words_cap, bs.desc.ctypes.data, bs.reads_cap, bs.heads.ctypes.data,
                                           bs.heads_cap, bs.head_off.ctypes.data, bs.spans.ctypes.data, C.byref(nw))
            T.append(("next", time.perf_counter() - t1))
            pool.put(bs)
            if n <= 0:
                break
        t0 = time.perf_counter(); lib.tps_reader_close(h); T.append(("close", time.perf_counter() - t0))
        tot = time.perf_counter() - t00
        t0 = time.perf_counter(); n2 = sum(pb.n for pb in (pb for pb in seqio.read_batches_packed(fq, pool)) if pb.release() is None); gen = time.perf_counter() - t0
        print("pass %d: " % rep + " ".join("%s=%.2f" % (k, v * 1e3) for k, v in T) + "  | calls total %.2f ms, generator total %.2f ms" % (tot * 1e3, gen * 1e3), flush=True)
finally:
    os.remove(fq)
