import os, time, mmap
src = "/tmp/cfr_src.bin"
with open(src, "wb") as h:
    h.write(os.urandom(1 << 20) * 300)
n = os.path.getsize(src)
def t_copy_file_range():
    fi = os.open(src, os.O_RDONLY); fo = os.open("/tmp/cfr_dst.bin", os.O_WRONLY | os.O_CREAT | os.O_TRUNC)
    t = time.perf_counter(); off = 0
    while off < n:
        got = os.copy_file_range(fi, fo, n - off, off, off); off += got
    dt = time.perf_counter() - t; os.close(fi); os.close(fo); return dt
def t_write_from_mmap():
    fi = os.open(src, os.O_RDONLY); m = mmap.mmap(fi, 0, prot=mmap.PROT_READ); fo = os.open("/tmp/cfr_dst2.bin", os.O_WRONLY | os.O_CREAT | os.O_TRUNC)
    mv = memoryview(m)
    t = time.perf_counter(); off = 0
    while off < n:
        off += os.write(fo, mv[off:off + (64 << 20)])
    dt = time.perf_counter() - t; mv.release(); m.close(); os.close(fi); os.close(fo); return dt
def t_sendfile():
    fi = os.open(src, os.O_RDONLY); fo = os.open("/tmp/cfr_dst3.bin", os.O_WRONLY | os.O_CREAT | os.O_TRUNC)
    t = time.perf_counter(); off = 0
    while off < n:
        off += os.sendfile(fo, fi, off, n - off)
    dt = time.perf_counter() - t; os.close(fi); os.close(fo); return dt
for name, f in (("copy_file_range", t_copy_file_range), ("write from mmap", t_write_from_mmap), ("sendfile", t_sendfile)):
    try:
        ts = [f() for _ in range(3)]
        print("%-18s %.3f s best  (%.1f GB/s)" % (name, min(ts), n / min(ts) / 1e9))
    except Exception as e:
        print(name, "failed", e)
for f in ("/tmp/cfr_src.bin", "/tmp/cfr_dst.bin", "/tmp/cfr_dst2.bin", "/tmp/cfr_dst3.bin"):
    try: os.unlink(f)
    except OSError: pass
