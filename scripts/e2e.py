"""End-to-end (PCIe- and parse-inclusive) rates for DESIGN.md: native parse, upload, scan."""
import gzip, os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from topsicle_amd import hiplib, synth, allsteps, seqio
motif = "CCCTAA"
pats = allsteps.patterns_to_search(motif, 4)
b, o, _ = synth.make_reads(10000, 15000, motif, 20250920)
sc = hiplib.HipScanner(0); sc.set_patterns(pats)
prm = hiplib.make_params(min_len=9000, min_count=allsteps.min_count_for_cutoff(0.7, 1000 / 6, 1000))
sc.upload(0, b, o); sc.scan(0, prm); sc.sync()
t0 = time.perf_counter(); sc.upload(0, b, o); t1 = time.perf_counter(); sc.scan(0, prm); sc.sync(); t2 = time.perf_counter()
print(f"upload (pageable numpy -> HBM, 150 MB): {(t1-t0)*1e3:.1f} ms = {b.size/(t1-t0)/1e9:.1f} GB/s ; scan {(t2-t1)*1e3:.2f} ms ; "
      f"upload+scan {b.size/(t2-t0)/1e9:.2f} G bases/s")
path = "/tmp/e2e_reads.fastq"
with open(path, "wb") as h:
    q = b"I" * 15000
    raw = b.tobytes()
    for i in range(10000):
        h.write(b"@r%d\n" % i + raw[o[i]:o[i+1]] + b"\n+\n" + q + b"\n")
t0 = time.perf_counter()
n = nb = 0
for rb in seqio.read_batches(path):
    sc.upload(0, rb.bases, rb.offsets); sc.scan(0, prm); sc.sync(); res = sc.results(0)
    n += len(rb); nb += int(rb.offsets[-1])
t1 = time.perf_counter()
print(f"plain FASTQ file -> native parse -> upload -> scan -> results: {n} reads, {nb/(t1-t0)/1e9:.2f} G bases/s ({(t1-t0)*1e3:.0f} ms)")
# the same through the batched pipeline (host decoding of batch i+1 overlaps the scan of batch i)
from topsicle_amd import batch
pool = batch.EnginePool([sc], pats)
for mb in (64 << 20, 32 << 20):
    t0 = time.perf_counter()
    n = nb = 0
    for rb, res, _s, _r, _w in pool.scan_file(path, prm, max_bases=mb):
        n += len(rb); nb += int(rb.offsets[-1])
    t1 = time.perf_counter()
    print(f"pipelined, {mb >> 20} MB batches: {n} reads, {nb/(t1-t0)/1e9:.2f} G bases/s ({(t1-t0)*1e3:.0f} ms)")
os.remove(path)
