"""Host I/O rates of the native reader (no GPU involved): plain FASTQ (mmap + thread team), bgzip'ed FASTQ (blocks
inflated in parallel), ordinary .gz (one zlib stream, one thread).  usage: io_bench.py [N_READS]"""
import gzip, os, struct, sys, time, zlib
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from topsicle_amd import seqio, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
b, o, _ = synth.make_reads(n, 15000, "CCCTAA", 3)
raw = b.tobytes()
rng = np.random.default_rng(1)
qual = bytes(rng.integers(40, 70, 15000, dtype=np.uint8))
text = b"".join(b"@read%d\n" % i + raw[o[i]:o[i + 1]] + b"\n+\n" + qual + b"\n" for i in range(n))
d = "/tmp/io_bench"
os.makedirs(d, exist_ok=True)
open(f"{d}/r.fastq", "wb").write(text)
with gzip.open(f"{d}/r_plain.fastq.gz", "wb", compresslevel=1) as h:
    h.write(text)
with open(f"{d}/r_bgzf.fastq.gz", "wb") as h:
    for lo in range(0, len(text), 65280):
        chunk = text[lo:lo + 65280]
        co = zlib.compressobj(1, zlib.DEFLATED, -15)
        body = co.compress(chunk) + co.flush()
        h.write(b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(body) + 8 - 1) +
                body + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
    h.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
for name in ("r.fastq", "r_bgzf.fastq.gz", "r_plain.fastq.gz"):
    best = 0
    for rep in range(2):
        t0 = time.perf_counter()
        nb = nr = 0
        for rb in seqio.read_batches(f"{d}/{name}"):
            nb += int(rb.offsets[-1]); nr += len(rb)
        dt = time.perf_counter() - t0
        best = max(best, nb / dt)
    assert nr == n and nb == len(raw), (name, nr, nb)
    print(f"{name:20s} {os.path.getsize(f'{d}/{name}') / 1e6:8.1f} MB on disk  {best / 1e9:6.2f} G bases/s")
for f in os.listdir(d):
    os.remove(f"{d}/{f}")
