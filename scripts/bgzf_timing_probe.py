"""Diagnostic (GPU box): the reader's timing lines (option "timing") over a BGZF file of config-2 reads with noisy quality lines."""
import os, sys, time
sys.path.insert(0, '.')
os.environ["TOPSICLE_IO_DEBUG"] = "timing"
import numpy as np
from topsicle_amd import e2e, seqio, synth
b, o, t = synth.make_reads(10000, 15000, "CCCTAA", seed=20250920, errors=synth.ONT)
e2e.write_fastq("/tmp/bgp.fastq", b, o, random_quality_seed=1)
e2e.write_bgzf("/tmp/bgp.fastq.gz", "/tmp/bgp.fastq")
pool = seqio.BufferPool(4, (64 << 20) // 16, 1 << 18)
for rep in range(3):
    t0 = time.perf_counter()
    n = 0
    for pb in seqio.read_batches_packed("/tmp/bgp.fastq.gz", pool):
        n += pb.n
        pb.release()
    print("pass", rep, n, "%.1f ms" % (1e3 * (time.perf_counter() - t0)), flush=True)
os.unlink("/tmp/bgp.fastq"); os.unlink("/tmp/bgp.fastq.gz")
