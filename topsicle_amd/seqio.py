"""Minimal FASTA / FASTQ (.gz) reader and writer for the host side.

Replaces the Biopython calls of the reference (Bio.SeqIO.parse / SeqIO.write,
allsteps.py:145, main.py:84-86) -- Biopython is not a dependency of this package.  Record ids
follow Biopython: the first whitespace-delimited token of the header line.
"""
from __future__ import annotations

import gzip
import io
import logging
from dataclasses import dataclass


@dataclass
class Record:
    id: str
    description: str      # full header line without the leading '>' / '@'
    seq: str
    qual: str | None = None

    def __len__(self):
        return len(self.seq)


def _open_text(path: str):
    if path.endswith(".gz"):
        return io.TextIOWrapper(gzip.open(path, "rb"), encoding="utf-8", newline="")
    return open(path, "rt", encoding="utf-8", newline="")


def check_file_type(filepath: str):
    """'fastq' / 'fasta' from the first character of the file, 0 if unknown (allsteps.py:36-50)."""
    try:
        with _open_text(filepath) as h:
            first = h.readline().strip()
    except Exception as e:  # same contract as the reference: log and return 0
        logging.error(f"Error checking file type: {e}")
        return 0
    if first.startswith("@"):
        return "fastq"
    if first.startswith(">"):
        return "fasta"
    logging.warning("Format cannot be identified. Check the input.")
    return 0


def _first_token(desc: str) -> str:
    parts = desc.split()
    return parts[0] if parts else ""


def parse(handle, fmt: str):
    if fmt == "fasta":
        desc, chunks = None, []
        for line in handle:
            line = line.rstrip("\r\n")
            if line.startswith(">"):
                if desc is not None:
                    yield Record(_first_token(desc), desc, "".join(chunks))
                desc, chunks = line[1:], []
            elif desc is not None:
                chunks.append(line.strip())
        if desc is not None:
            yield Record(_first_token(desc), desc, "".join(chunks))
    elif fmt == "fastq":
        while True:
            head = handle.readline()
            if not head:
                return
            head = head.rstrip("\r\n")
            if not head:
                continue
            if not head.startswith("@"):
                raise ValueError("FASTQ record does not start with '@'")
            seq_lines = []
            line = handle.readline()
            while line and not line.startswith("+"):
                seq_lines.append(line.strip())
                line = handle.readline()
            seq = "".join(seq_lines)
            qual = ""
            while len(qual) < len(seq):
                q = handle.readline()
                if not q:
                    break
                qual += q.rstrip("\r\n")
            if len(qual) != len(seq):
                # a quality string that is too short swallowed the next header; one that is too long leaves its tail
                # where a header should be: either way the file is mis-framed from here on (Biopython raises as well)
                raise ValueError(f"FASTQ record {head[1:].split()[0] if head[1:].split() else ''!r}: "
                                 f"{len(qual)} quality characters for {len(seq)} bases")
            desc = head[1:]
            yield Record(_first_token(desc), desc, seq, qual)
    else:
        raise ValueError(f"unknown format {fmt!r}")


def read_records(filepath: str):
    """Generator over the records of a FASTA/FASTQ(.gz) file (unzip_file, allsteps.py:127-149).
    Parse errors are logged and end the iteration, as upstream."""
    if not isinstance(filepath, str):
        logging.error("Input must be a string representing the file path.")
        return
    fmt = check_file_type(filepath)
    if not fmt:
        logging.error("File type could not be determined or is unsupported.")
        return
    try:
        with _open_text(filepath) as h:
            yield from parse(h, fmt)
    except Exception as e:
        logging.error(f"Error parsing file: {e}")


def write_record(handle, rec: Record, fmt: str):
    """One record in the layout Biopython's SeqIO.write produces (FASTA wrapped at 60)."""
    if fmt == "fastq":
        if rec.qual is None:
            raise ValueError("no qualities for FASTQ output")
        handle.write(f"@{rec.description}\n{rec.seq}\n+\n{rec.qual}\n")
    elif fmt == "fasta":
        handle.write(f">{rec.description}\n")
        for i in range(0, len(rec.seq), 60):
            handle.write(rec.seq[i:i + 60] + "\n")
    else:
        raise ValueError(fmt)


# ---------------------------------------------------------------------------- native batch reader
class RecordBatch:
    """A batch of reads in the layout tps_batch_upload takes (concatenated bases + offsets); record
    objects are only materialised on request (for the few reads that are written back out)."""

    def __init__(self, bases, offsets, heads: bytes, head_off, quals=None, fmt="fastq"):
        self.bases, self.offsets, self.heads, self.head_off, self.quals, self.fmt = bases, offsets, heads, head_off, quals, fmt
        self.n = len(offsets) - 1
        self._ids = None

    def __len__(self):
        return self.n

    @property
    def ids(self):
        if self._ids is None:
            h, o = self.heads, self.head_off
            self._ids = [_first_token(h[o[i]:o[i + 1]].decode("utf-8", "replace")) for i in range(self.n)]
        return self._ids

    def record(self, i: int) -> Record:
        d = self.heads[self.head_off[i]:self.head_off[i + 1]].decode("utf-8", "replace")
        lo, hi = int(self.offsets[i]), int(self.offsets[i + 1])
        seq = self.bases[lo:hi].tobytes().decode("ascii", "replace")
        qual = self.quals[lo:hi].tobytes().decode("ascii", "replace") if self.quals is not None else None
        return Record(_first_token(d), d, seq, qual)

    @classmethod
    def from_records(cls, recs, fmt="fastq"):
        import numpy as np
        heads = [r.description.encode() for r in recs]
        seqs = [r.seq.encode("ascii", "replace") for r in recs]
        offsets = np.zeros(len(recs) + 1, np.int64)
        head_off = np.zeros(len(recs) + 1, np.int64)
        if recs:
            np.cumsum([len(x) for x in seqs], out=offsets[1:])
            np.cumsum([len(x) for x in heads], out=head_off[1:])
        bases = np.frombuffer(b"".join(seqs), dtype=np.uint8) if recs else np.zeros(0, np.uint8)
        quals = None
        if recs and all(r.qual is not None for r in recs):
            quals = np.frombuffer(b"".join((r.qual + "!" * len(r.seq))[:len(r.seq)].encode() for r in recs), dtype=np.uint8)
        b = cls(bases, offsets, b"".join(heads), head_off, quals, fmt)
        b._ids = [r.id for r in recs]
        return b


_io_lib = None

# libtopsicle_io.so's exports (include/topsicle_io.h), bound from this one table: name -> (restype, argtypes) as ctypes type names
IO_EXPORTS = {
    "tps_io_last_error": ("c_char_p", []),
    "tps_reader_open": ("c_int", ["c_char_p", "POINTER(c_void_p)"]),
    "tps_reader_open_range": ("c_int", ["c_char_p", "c_int64", "c_int64", "c_int32", "POINTER(c_void_p)"]),
    "tps_reader_range_info": ("c_int", ["c_void_p", "POINTER(c_int64)", "POINTER(c_int64)"]),
    "tps_reader_format": ("c_int", ["c_void_p"]),
    "tps_reader_close": (None, ["c_void_p"]),
    "tps_reader_next": ("c_int64", ["c_void_p", "c_void_p", "c_int64", "c_void_p", "c_int64", "c_void_p", "c_int64", "c_void_p", "c_void_p"]),
    "tps_reader_next_packed": ("c_int64", ["c_void_p", "c_void_p", "c_void_p", "c_int64", "c_void_p", "c_int64", "c_void_p", "c_int64", "c_void_p",
                                           "c_void_p", "POINTER(c_int64)"]),
    "tps_reader_next_heads": ("c_int64", ["c_void_p", "c_int32", "c_void_p", "c_void_p", "c_int64", "c_void_p", "c_int64", "c_void_p", "c_int64",
                                          "c_void_p", "c_void_p", "c_void_p", "POINTER(c_int64)"]),
    "tps_reader_text_hold": ("c_int", ["c_void_p", "POINTER(c_void_p)", "POINTER(c_int64)", "POINTER(c_void_p)"]),
    "tps_text_release": (None, ["c_void_p"]),
    "tps_pack_spans": ("c_int64", ["c_void_p", "c_int64", "c_int32", "c_void_p", "c_void_p", "c_void_p", "c_void_p", "c_int64", "c_int32", "c_void_p",
                                   "c_void_p", "c_void_p", "c_int64"]),
    "tps_packed_words_total": ("c_int64", ["c_void_p", "c_int64"]),
    "tps_pack_reads": ("c_int64", ["c_void_p", "c_void_p", "c_int64", "c_void_p", "c_void_p", "c_void_p", "c_int32"]),
    "tps_write_fastq_spans": ("c_int64", ["c_int", "c_void_p", "c_int64", "c_void_p", "c_void_p", "c_void_p", "c_int64"]),
    "tps_write_fastq_spans_at": ("c_int64", ["c_int", "c_int64", "c_void_p", "c_int64", "c_void_p", "c_void_p", "c_void_p", "c_int64"]),
    "tps_fastq_spans_bytes": ("c_int64", ["c_void_p", "c_void_p", "c_void_p", "c_int64"]),
    "tps_crc32": ("c_uint32", ["c_uint32", "c_void_p", "c_int64"]),
    "tps_crc32_combine": ("c_uint32", ["c_uint32", "c_uint32", "c_int64"]),
    "tps_io_set_option": ("c_int", ["c_char_p", "c_int64"]),
    "tps_gz_inflate": ("c_int64", ["c_char_p", "c_void_p", "c_int64", "c_int32", "c_int64", "c_void_p"]),
}


def _load_io():
    """libtopsicle_io.so (csrc/tps_io.cpp): C++ FASTA/FASTQ(.gz) decoder; None if not built."""
    global _io_lib
    if _io_lib is None:
        import ctypes as C
        import os
        # TOPSICLE_IO_LIB: another build of csrc/tps_io.cpp (the sanitizer build of tests/test_sanitizers.py)
        path = os.environ.get("TOPSICLE_IO_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libtopsicle_io.so")
        if not os.path.exists(path):
            _io_lib = False
            return None
        lib = C.CDLL(path)
        ns = {"POINTER": C.POINTER, **{n: getattr(C, n) for n in ("c_char_p", "c_void_p", "c_int", "c_int32", "c_int64", "c_uint32")}}
        for name, (res, args) in IO_EXPORTS.items():
            fn = getattr(lib, name)
            fn.restype = None if res is None else eval(res, ns)
            fn.argtypes = [eval(a, ns) for a in args]
        # $TOPSICLE_IO_DEBUG = "key=value,key": the one environment variable through which diagnostics reach the reader (the library
        # itself reads no environment: tps_io_set_option); unset in normal use
        for item in os.environ.get("TOPSICLE_IO_DEBUG", "").split(","):
            item = item.strip()
            if item:
                key, _, val = item.partition("=")
                if lib.tps_io_set_option(key.strip().encode(), int(val) if val.strip() else 1) != 0:
                    raise ValueError("TOPSICLE_IO_DEBUG: " + lib.tps_io_last_error().decode())
        _io_lib = lib
    return _io_lib or None


IO_OPTION_DEFAULTS = {"threads": 0, "timing": 0, "bgzf_group": 0, "pack_min_span": -1, "no_pargz": 0, "pargz_min": -1}


def io_option(key: str, value: int):
    """tps_io_set_option: tests and diagnostics (process-wide; IO_OPTION_DEFAULTS restores)."""
    lib = _load_io()
    if lib is None:
        raise RuntimeError("libtopsicle_io.so is not built")
    if lib.tps_io_set_option(key.encode(), int(value)) != 0:
        raise ValueError(lib.tps_io_last_error().decode())


def read_batches(filepath: str, max_bases: int = 256 << 20, max_records: int = 1 << 20, want_quals: bool = True):
    """Generator of RecordBatch over a FASTA/FASTQ(.gz) file: the native decoder when it is built,
    otherwise the pure-Python parser above (same records either way; this is host I/O, not compute)."""
    import numpy as np
    lib = _load_io()
    if lib is None:
        fmt = check_file_type(filepath)
        cur, nb = [], 0
        for rec in read_records(filepath):
            if cur and (nb + len(rec.seq) > max_bases or len(cur) >= max_records):
                yield RecordBatch.from_records(cur, fmt or "fasta")
                cur, nb = [], 0
            cur.append(rec)
            nb += len(rec.seq)
        if cur:
            yield RecordBatch.from_records(cur, fmt or "fasta")
        return
    import ctypes as C
    h = C.c_void_p()
    if lib.tps_reader_open(filepath.encode(), C.byref(h)) != 0:
        logging.error("Error parsing file: %s", lib.tps_io_last_error().decode())
        return
    try:
        fmt = {1: "fasta", 2: "fastq"}.get(lib.tps_reader_format(h), "fasta")
        cap = max_bases
        while True:
            bases = np.empty(cap, np.uint8)
            quals = np.empty(cap, np.uint8) if (want_quals and fmt == "fastq") else None
            nrec = min(max_records, cap // 64 + 1024)
            offsets = np.empty(nrec + 1, np.int64)
            head_off = np.empty(nrec + 1, np.int64)
            heads = np.empty(max(cap // 8, 1 << 20), np.uint8)
            n = lib.tps_reader_next(h, bases.ctypes.data, cap, offsets.ctypes.data, nrec, heads.ctypes.data, len(heads),
                                    head_off.ctypes.data, quals.ctypes.data if quals is not None else None)
            if n == -2:                     # a single record larger than the buffers: grow and retry
                cap *= 2
                continue
            if n < 0:
                logging.error("Error parsing file: %s", lib.tps_io_last_error().decode())
                return
            if n == 0:
                return
            nb = int(offsets[n])
            yield RecordBatch(bases[:nb], offsets[:n + 1].copy(), heads[:int(head_off[n])].tobytes(), head_off[:n + 1].copy(),
                              quals[:nb] if quals is not None else None, fmt)
    finally:
        lib.tps_reader_close(h)


# ---------------------------------------------------------------------------- packed batches (upload format)
def pack_reads_host(bases, offsets, out=None, threads: int = 0):
    """ASCII batch -> the packed upload format of include/topsicle_hip.h (2 bits per base + invalid mask + read
    descriptors) with the native thread team.  `out` = (seq2, inv, desc) buffers to fill (e.g. pinned); returns
    (seq2[:n_words], inv[:n_words], desc[:n])."""
    import ctypes as C
    import numpy as np
    from . import hiplib
    lib = _load_io()
    if lib is None:
        raise RuntimeError("libtopsicle_io.so is not built")
    bases = np.ascontiguousarray(bases, np.uint8)
    offsets = np.ascontiguousarray(offsets, np.int64)
    n = len(offsets) - 1
    nw = int(lib.tps_packed_words_total(offsets.ctypes.data, n))
    if out is None:
        out = (np.empty(max(nw, 4), np.uint32), np.empty(max(nw, 4), np.uint16), np.empty(max(n, 1), hiplib.DESC_DTYPE))
    seq2, inv, desc = out
    if len(seq2) < nw or len(inv) < nw or len(desc) < n:
        raise ValueError("packed buffers too small")
    got = lib.tps_pack_reads(bases.ctypes.data, offsets.ctypes.data, n, seq2.ctypes.data, inv.ctypes.data, desc.ctypes.data, threads)
    if got < 0:
        raise RuntimeError(lib.tps_io_last_error().decode())
    return seq2[:nw], inv[:nw], desc[:n]


class BufferSet:
    """One set of upload staging buffers (seq2, inv, desc), typically pinned (HipScanner.host_alloc)."""

    def __init__(self, words_cap: int, reads_cap: int, alloc=None):
        import numpy as np
        from . import hiplib
        self.words_cap, self.reads_cap = int(words_cap), int(reads_cap)
        if alloc is None:
            def alloc(nbytes):
                return np.empty(nbytes, np.uint8)
        self.seq2 = alloc(4 * self.words_cap).view(np.uint32)
        self.inv = alloc(2 * self.words_cap).view(np.uint16)
        self.desc = alloc(16 * self.reads_cap).view(hiplib.DESC_DTYPE)
        # scratch the packed reader fills per batch (the used parts are copied into the PackedBatch: they outlive release())
        self.heads_cap = max(self.words_cap * 2, 1 << 20)
        self.heads = np.empty(self.heads_cap, np.uint8)
        self.head_off = np.empty(self.reads_cap + 1, np.int64)
        self.spans = np.empty((self.reads_cap, 4), np.int64)
        self.full_len = np.empty(self.reads_cap, np.int32)          # (heads mode: every read's own length)


class BufferPool:
    """A few BufferSets handed round between the reader thread and the upload workers (double buffering: the reader fills
    one set while others are in flight)."""

    def __init__(self, n_sets: int, words_cap: int, reads_cap: int, alloc=None):
        import queue
        self.words_cap, self.reads_cap = int(words_cap), int(reads_cap)
        self._free = queue.Queue()
        self._alloc = alloc
        self.n_sets = int(n_sets)
        self._made = 0                                # sets are allocated on demand: a file of two batches pins two sets, not n_sets
        import threading
        self.abort = threading.Event()
        self._lock = threading.Lock()

    def get(self) -> BufferSet:
        """A free set; while none is free the `abort` event is polled, so that a reader whose consumer has gone away (an error
        downstream, an abandoned generator) ends instead of waiting for a buffer that will never come back."""
        import queue
        try:
            return self._free.get_nowait()
        except queue.Empty:
            pass
        with self._lock:
            make = self._made < self.n_sets
            if make:
                self._made += 1
        if make:
            return BufferSet(self.words_cap, self.reads_cap, self._alloc)
        while True:
            try:
                return self._free.get(timeout=0.2)
            except queue.Empty:
                if self.abort.is_set():
                    raise RuntimeError("the batch pipeline was stopped")

    def put(self, bs: BufferSet):
        self._free.put(bs)


class PackedBatch:
    """A batch of reads in the packed upload format, plus what is needed to write single records back out: either the
    records' spans in the mmap'ed input file (plain FASTQ, nothing was copied) or the ASCII RecordBatch it was packed from."""

    def __init__(self, seq2, inv, desc, heads, head_off, fmt, spans=None, text=None, ascii_batch=None, bufset=None, pool=None, full_len=None):
        self.seq2, self.inv, self.desc, self.heads, self.head_off, self.fmt = seq2, inv, desc, heads, head_off, fmt
        self.spans, self.text, self.ascii_batch = spans, text, ascii_batch
        # heads mode (read_batches_packed(heads_bp=...)): seq2 / desc hold every read's first + last heads_bp bases only (all that
        # step 1 looks at); full_len = the reads' own lengths.  batch.scan_jobs runs step 1 on them and packs the scanned part of the
        # reads that pass from their spans (pack_spans).  After release() desc["len"] is the read's own length, as for any batch.
        self.full_len = full_len
        self.n = len(desc)
        self.n_bases = int((desc["len"] if full_len is None else full_len).sum(dtype="int64")) if self.n else 0
        self._bufset, self._pool = bufset, pool
        self._ids = None

    def __len__(self):
        return self.n

    @property
    def any_invalid(self) -> bool:
        return bool((self.desc["flags"] & 1).any())

    def release(self):
        """Hand the staging buffers back (after the upload has completed).  seq2 / inv / desc must not be used afterwards."""
        if self._pool is not None and self._bufset is not None:
            import numpy as np
            self.desc = np.array(self.desc)               # lengths stay available for the writers
            if self.full_len is not None:
                self.desc["len"] = self.full_len          # (heads mode: from here on the batch describes the whole reads)
            self.seq2 = self.inv = None
            self._pool.put(self._bufset)
            self._bufset = None

    def head(self, i: int) -> str:
        return bytes(self.heads[self.head_off[i]:self.head_off[i + 1]]).decode("utf-8", "replace")

    @property
    def ids(self):
        if self._ids is None:
            self._ids = [_first_token(self.head(i)) for i in range(self.n)]
        return self._ids

    def read_id(self, i: int) -> str:
        return self._ids[i] if self._ids is not None else _first_token(self.head(i))

    def read_len(self, i: int) -> int:
        return int(self.desc["len"][i] if self.full_len is None else self.full_len[i])

    def seq_bytes(self, i: int) -> bytes:
        if self.spans is not None:
            s0, n = int(self.spans[i, 2]), self.read_len(i)
            if self.fmt == "fasta":
                # FASTA spans: [first base, end of the record's sequence text); a wrapped sequence is joined here (the few records
                # that are written back out), exactly as the reader joined it for packing: line ends dropped, nothing else
                raw = self.text[s0:int(self.spans[i, 3])]
                head = raw[:n]
                if b"\n" in head or b"\r" in head:    # wrapped (a sequence on one line has its n bases in front of the first line end)
                    raw = b"".join(ln[:-1] if ln.endswith(b"\r") else ln for ln in raw.split(b"\n"))
                return raw[:n]
            return self._joined(s0, n)
        b = self.ascii_batch
        return b.bases[int(b.offsets[i]):int(b.offsets[i + 1])].tobytes()

    def _joined(self, off: int, n: int) -> bytes:
        """n sequence / quality characters of a FASTQ record from text offset `off` on: the n bytes themselves, or -- a multi-line
        record (the reader joined it the same way for packing) -- the first n bytes that are not line ends."""
        raw = self.text[off:off + n]
        if b"\n" not in raw and b"\r" not in raw:
            return raw
        raw = self.text[off:off + 3 * n + 16]         # (at worst one character per CRLF line)
        return raw.replace(b"\r", b"").replace(b"\n", b"")[:n]

    def qual_bytes(self, i: int):
        if self.fmt == "fasta":
            return None                       # (a FASTA span's fourth entry is the end of the sequence text, not a quality offset)
        if self.spans is not None:
            return self._joined(int(self.spans[i, 3]), self.read_len(i))
        b = self.ascii_batch
        return None if b.quals is None else b.quals[int(b.offsets[i]):int(b.offsets[i + 1])].tobytes()

    def record(self, i: int) -> Record:
        d = self.head(i)
        q = self.qual_bytes(i)
        return Record(_first_token(d), d, self.seq_bytes(i).decode("ascii", "replace"), None if q is None else q.decode("ascii", "replace"))

    def _drop_text_pages(self, text_arr):
        """After this batch's records have been written from the MAPPED input file: its stretch of the mapping leaves the resident
        set (the page cache keeps the data; nothing reads this batch's text again) -- a 3 GB plain input no longer sits in RSS."""
        import mmap
        if not isinstance(self.text, mmap.mmap) or self.spans is None or not len(self.spans) or not hasattr(mmap, "MADV_DONTNEED"):
            return
        lo = int(self.spans[0][0]) & ~(mmap.PAGESIZE - 1)
        hi = int(self.spans[-1][0]) & ~(mmap.PAGESIZE - 1)              # (up to the last record's header: the next batch may share that page)
        if hi > lo:
            try:
                self.text.madvise(mmap.MADV_DONTNEED, lo, hi - lo)
            except (OSError, ValueError):
                pass

    def native_fastq_bytes(self, indices, fmt: str):
        """Bytes the native writer puts out for these records (header + 2 x bases + 6 each), or None when this batch does not leave
        through it (FASTA, ASCII batches): the caller can then give every batch its place in the file up front and let several
        threads write at once (write_records(..., offset))."""
        import numpy as np
        if not (fmt == "fastq" and self.fmt == "fastq" and self.spans is not None and self.text is not None) or _load_io() is None:
            return None
        idx = np.ascontiguousarray(indices, dtype=np.int64)
        lens = (self.desc["len"] if self.full_len is None else self.full_len).astype(np.int64)
        return int((np.asarray(self.spans)[idx, 1] + 2 * lens[idx] + 6).sum())

    def write_records(self, handle, indices, fmt: str, offset=None):
        """Write the given records to a BINARY handle in the layout Biopython's SeqIO.write produces (main.py:84-86).  Records
        of a batch that was packed straight from a mmap'ed plain FASTQ file leave through the native writer: writev from the
        mapping, no copy in user space (tps_write_fastq_spans).  offset (native path only: native_fastq_bytes is not None): write
        at that byte of the file whatever the handle's position (pwritev) -- several threads, one file."""
        import numpy as np
        if fmt == "fastq" and self.fmt == "fastq" and self.spans is not None and self.text is not None and len(indices):
            lib = _load_io()
            try:
                fd = handle.fileno()                  # (an in-memory handle has none: the Python loop below serves it)
            except (AttributeError, OSError, ValueError):
                fd = None
            if lib is not None and fd is not None:
                handle.flush()
                text = np.frombuffer(self.text.buffer() if hasattr(self.text, "buffer") else self.text, dtype=np.uint8)
                idx = np.ascontiguousarray(indices, dtype=np.int64)
                lens = np.ascontiguousarray(self.desc["len"] if self.full_len is None else self.full_len, dtype=np.int32)
                spans = np.ascontiguousarray(self.spans, dtype=np.int64)
                if offset is None:
                    got = lib.tps_write_fastq_spans(fd, text.ctypes.data, len(text), spans.ctypes.data, lens.ctypes.data, idx.ctypes.data, len(idx))
                else:
                    got = lib.tps_write_fastq_spans_at(fd, int(offset), text.ctypes.data, len(text), spans.ctypes.data, lens.ctypes.data, idx.ctypes.data, len(idx))
                if got < 0:
                    raise OSError(lib.tps_io_last_error().decode())
                self._drop_text_pages(text)
                return
        if offset is not None:
            raise ValueError("offset: only for batches the native writer takes (native_fastq_bytes)")
        out = []
        for i in indices:
            i = int(i)
            head = bytes(self.heads[self.head_off[i]:self.head_off[i + 1]])
            seq = self.seq_bytes(i)
            if fmt == "fastq":
                q = self.qual_bytes(i)
                if q is None:
                    raise ValueError("no qualities for FASTQ output")
                out += [b"@", head, b"\n", seq, b"\n+\n", q, b"\n"]
            else:
                out += [b">", head, b"\n"]
                out += [seq[j:j + 60] + b"\n" for j in range(0, len(seq), 60)]
            if len(out) > 4096:
                handle.write(b"".join(out))
                out = []
        if out:
            handle.write(b"".join(out))


class _HeldText:
    """A window of inflated text that belongs to the native reader (compressed input), kept alive by a reference
    (tps_reader_text_hold) until this object goes: a read-only buffer like the mmap of a plain file."""

    def __init__(self, lib, ptr, length, hold):
        import ctypes as C
        self._lib, self._hold = lib, hold
        self._arr = (C.c_char * length).from_address(ptr) if length else (C.c_char * 0)()
        self._mv = memoryview(self._arr).cast("B")

    def __len__(self):
        return len(self._mv)

    def __getitem__(self, key):
        v = self._mv[key]
        return bytes(v) if isinstance(key, slice) else v

    def __buffer__(self, flags):              # (Python >= 3.12)
        return self._mv

    def buffer(self):
        return self._mv

    def __del__(self):
        try:
            hold, self._hold = self._hold, None
            if hold:
                self._mv.release()
                self._lib.tps_text_release(hold)
        except Exception:
            pass


def pack_spans(pb: "PackedBatch", idx, tails, maxlen: int):
    """Second pass of the heads mode: (seq2, inv, desc) of the part of reads idx[j] of `pb` a scan of tail tails[j] touches -- the
    first (tail 0) / last (tail 1) min(length, maxlen) bases, each as a read of its own -- packed by the native team from the
    batch's text (tps_pack_spans).  Plain arrays (a few reads: the ones that passed step 1)."""
    import ctypes as C
    import numpy as np
    from . import hiplib
    lib = _load_io()
    idx = np.ascontiguousarray(idx, np.int64)
    tails = np.ascontiguousarray(tails, np.uint8)
    full = np.ascontiguousarray(pb.full_len if pb.full_len is not None else pb.desc["len"], np.int32)
    m = np.minimum(full[idx].astype(np.int64), int(maxlen))
    words = int((((m + 63) // 64) * 4).sum())
    seq2 = np.zeros(max(words, 4), np.uint32)
    inv = np.zeros(max(words, 4), np.uint16)
    desc = np.zeros(len(idx), hiplib.DESC_DTYPE)
    text = np.frombuffer(pb.text.buffer() if hasattr(pb.text, "buffer") else pb.text, dtype=np.uint8)
    spans = np.ascontiguousarray(pb.spans, np.int64)
    got = lib.tps_pack_spans(text.ctypes.data, len(text), 1 if pb.fmt == "fasta" else 0, spans.ctypes.data, full.ctypes.data, idx.ctypes.data,
                             tails.ctypes.data, len(idx), int(maxlen), seq2.ctypes.data, inv.ctypes.data, desc.ctypes.data, len(seq2))
    if got < 0:
        raise RuntimeError(lib.tps_io_last_error().decode())
    return seq2[:got], inv[:got], desc


def shard_ranges(filepath: str, n_shards: int, min_bytes: int = 64 << 20):
    """Byte ranges [(lo, hi), ...] that cut a plain FASTA / FASTQ file -- or a BGZF-compressed one, at block boundaries -- into at most
    n_shards readers of at least min_bytes each (read_batches_packed(byte_range=...): a record belongs to the range its first byte
    lies in; BGZF: to the reader whose blocks hold the line end in front of it), or None when the file cannot be cut: ordinary gzip
    (one deflate stream), too small, unreadable."""
    import os
    try:
        size = os.path.getsize(filepath)
        with open(filepath, "rb") as fh:
            magic = fh.read(2)
    except OSError:
        return None
    if n_shards < 2:
        return None
    if magic == b"\x1f\x8b":
        # BGZF (bgzip): independent blocks -- the ranges are ranges of COMPRESSED bytes, a reader owns the blocks that start in its
        # range; an ordinary gzip file is one deflate stream: one reader
        try:
            with open(filepath, "rb") as fh:
                head = fh.read(18)
        except OSError:
            return None
        if len(head) < 18 or head[2] != 8 or not (head[3] & 4) or head[12:14] != b"BC":
            return None
    n = min(int(n_shards), max(1, size // max(int(min_bytes), 1)))
    if n < 2:
        return None
    cuts = [size * i // n for i in range(n + 1)]
    return list(zip(cuts[:-1], cuts[1:]))


def read_batches_packed(filepath: str, pool: BufferPool, max_records: int = 1 << 20, heads_bp=0, first_batch_records: int = 0,
                        byte_range=None, threads: int = 0, range_info=None):
    """Generator of PackedBatch over a FASTA/FASTQ(.gz) file.  Plain FASTQ is packed straight from the mmap'ed file by
    the native thread team (no ASCII copy, qualities untouched); other inputs are decoded to ASCII batches first
    (read_batches) and packed by the same team.  Every batch owns one BufferSet of `pool` until `release()`.
    heads_bp (an int, or a callable asked before every batch): > 0 = heads mode, see PackedBatch.full_len.
    first_batch_records > 0: the first batch holds at most that many records (a small probe: batch.EnginePool's auto mode).
    byte_range = (lo, hi): only the records that START in that byte range of a plain file, decoded by a team of `threads` threads
    (tps_reader_open_range: one reader per shard of one big file); range_info (a dict) receives "first" / "stopped" when the range is
    exhausted -- the caller checks the seams (shard i stopped where shard i + 1 began)."""
    import ctypes as C
    import mmap
    import numpy as np
    lib = _load_io()
    if lib is None:
        raise RuntimeError("libtopsicle_io.so is not built (python -c 'import __graft_entry__ as g; g.build()')")
    h = C.c_void_p()
    if byte_range is not None:
        if lib.tps_reader_open_range(filepath.encode(), int(byte_range[0]), int(byte_range[1]), int(threads), C.byref(h)) != 0:
            raise RuntimeError("cannot read %s by byte ranges: %s" % (filepath, lib.tps_io_last_error().decode()))
    elif lib.tps_reader_open(filepath.encode(), C.byref(h)) != 0:
        logging.error("Error parsing file: %s", lib.tps_io_last_error().decode())
        return
    fh = mm = None
    try:
        fmt = {1: "fasta", 2: "fastq"}.get(lib.tps_reader_format(h), "fasta")
        packed_mode = True
        nrec_cap = min(max_records, pool.reads_cap)
        if first_batch_records > 0:
            nrec_full, nrec_cap = nrec_cap, min(nrec_cap, int(first_batch_records))
        while packed_mode:
            bs = pool.get()
            heads, head_off, spans = bs.heads, bs.head_off, bs.spans
            nw = C.c_int64(0)
            hb = int(heads_bp() if callable(heads_bp) else heads_bp)
            if hb > 0:
                n = lib.tps_reader_next_heads(h, hb, bs.seq2.ctypes.data, bs.inv.ctypes.data, bs.words_cap, bs.desc.ctypes.data, nrec_cap,
                                              heads.ctypes.data, bs.heads_cap, head_off.ctypes.data, spans.ctypes.data, bs.full_len.ctypes.data,
                                              C.byref(nw))
            else:
                n = lib.tps_reader_next_packed(h, bs.seq2.ctypes.data, bs.inv.ctypes.data, bs.words_cap, bs.desc.ctypes.data, nrec_cap,
                                               heads.ctypes.data, bs.heads_cap, head_off.ctypes.data, spans.ctypes.data, C.byref(nw))
            if n > 0:
                # the text the spans point into: the mapped file itself, or -- compressed input -- the reader's window of inflated
                # text, which this batch keeps alive (tps_reader_text_hold) until it is gone
                tp, tl, th = C.c_void_p(), C.c_int64(0), C.c_void_p()
                lib.tps_reader_text_hold(h, C.byref(tp), C.byref(tl), C.byref(th))
                if th.value:
                    text = _HeldText(lib, tp.value, tl.value, th.value)
                else:
                    if mm is None:
                        fh = open(filepath, "rb")
                        mm = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ)
                    text = mm
                yield PackedBatch(bs.seq2[:nw.value], bs.inv[:nw.value], bs.desc[:n], heads[:int(head_off[n])].copy(), head_off[:n + 1].copy(),
                                  fmt, spans=spans[:n].copy(), text=text, bufset=bs, pool=pool,
                                  full_len=bs.full_len[:n].copy() if hb > 0 else None)
                if first_batch_records > 0:
                    nrec_cap, first_batch_records = nrec_full, 0
                continue
            pool.put(bs)
            if n == 0:
                return
            if n == -4:
                packed_mode = False          # odd records: ASCII batches from here on
                logging.info("%s: records the thread-team decoder does not take (blank or padded lines inside a record, lone CRs): "
                             "the one-thread streaming decoder reads the rest", filepath)
            elif n == -2:
                raise RuntimeError("a single read does not fit the upload buffers; raise the batch size")
            else:
                logging.error("Error parsing file: %s", lib.tps_io_last_error().decode())
                return
        # ASCII path: decode with tps_reader_next into plain arrays, then pack into a pool buffer
        cap = pool.words_cap * 16 - 64 * nrec_cap if pool.words_cap * 16 > 128 * nrec_cap else pool.words_cap * 8
        while True:
            bases = np.empty(cap, np.uint8)
            quals = np.empty(cap, np.uint8) if fmt == "fastq" else None
            nrec = min(nrec_cap, cap // 64 + 1024)
            offsets = np.empty(nrec + 1, np.int64)
            head_off = np.empty(nrec + 1, np.int64)
            heads = np.empty(max(cap // 8, 1 << 20), np.uint8)
            n = lib.tps_reader_next(h, bases.ctypes.data, cap, offsets.ctypes.data, nrec, heads.ctypes.data, len(heads),
                                    head_off.ctypes.data, quals.ctypes.data if quals is not None else None)
            if n == -2:
                raise RuntimeError("a single read does not fit the upload buffers; raise the batch size")
            if n < 0:
                logging.error("Error parsing file: %s", lib.tps_io_last_error().decode())
                return
            if n == 0:
                return
            nb = int(offsets[n])
            rb = RecordBatch(bases[:nb], offsets[:n + 1].copy(), heads[:int(head_off[n])].tobytes(), head_off[:n + 1].copy(),
                             quals[:nb] if quals is not None else None, fmt)
            bs = pool.get()
            seq2, inv, desc = pack_reads_host(rb.bases, rb.offsets, out=(bs.seq2, bs.inv, bs.desc))
            yield PackedBatch(seq2, inv, desc, np.frombuffer(rb.heads, np.uint8), rb.head_off, fmt, ascii_batch=rb, bufset=bs, pool=pool)
    finally:
        if range_info is not None and byte_range is not None:
            a, b = C.c_int64(0), C.c_int64(0)
            if lib.tps_reader_range_info(h, C.byref(a), C.byref(b)) == 0:
                range_info["first"], range_info["stopped"] = a.value, b.value
        lib.tps_reader_close(h)
        # (the mapping stays alive as long as a yielded batch references it)
