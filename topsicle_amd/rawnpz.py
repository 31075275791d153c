"""`--rawcountformat npz`: the per-window, per-pattern counts of one input file and one k as ONE columnar archive, written
once and in place.

The reference writes one `rawcount_{k}_{i}.csv` per passing read (Topsicle/main.py:146-150; the rows are built window by
window in Topsicle/allsteps.py:398-411): 1.2 MB of text per read, which bounds the raw-count workload end to end (SURVEY
section 8 f2).  The archive holds the same numbers: counts[win_off[i]:win_off[i + 1], p] is read i's count of pattern p in the
window that starts at position (w - win_off[i]) * slide.

Round 5: the rows go from the device to their final place in the file ONCE.  The archive is an ordinary stored (uncompressed)
zip -- `np.load` reads it like any .npz -- whose first member is `counts.npy`, laid out by hand:

    0      zip local header of counts.npy (zip64 sizes)
    60     the .npy header, padded with blanks so that
    4096   the rows start on a page boundary; batch after batch, in file order
    ...    read_id / tail / win_off / pattern / slide as small members, the central directory

The worker thread of a batch claims the batch's byte range as soon as its scan has said which reads pass (claims are handed
out in batch order, so the rows are in file order although batches finish out of order) and the library copies the rows
device -> pinned piece -> pwritev (`tps_batch_raw_to_fd`), checksumming each piece on the way; at the end the per-batch CRCs
are joined with crc32_combine and the two headers are written.  Round 4 spooled the rows to a side file from a Python loop per
read and copied the spool into the archive afterwards: the rows crossed memory four times on one thread (8.4 s for BASELINE
configs[4]'s per-GPU shard, of which the scans are 3 ms a batch).
"""
from __future__ import annotations

import os
import struct
import threading
import zlib

import numpy as np

DATA_OFF = 4096                     # counts.npy's rows start here
_NAME = b"counts.npy"
_LOCAL_LEN = 30 + len(_NAME) + 20   # local header + name + zip64 extra
_DOS_TIME, _DOS_DATE = 0, (1 << 5) | 1          # 1980-01-01, like numpy's own archives when no time is known


def _crc32_combine(crc1: int, crc2: int, len2: int) -> int:
    """zlib's crc32_combine (GF(2) matrix squaring); the native one (libtopsicle_io.so: tps_crc32_combine) when it is there."""
    from . import seqio
    lib = seqio._load_io()
    if lib is not None and hasattr(lib, "tps_crc32_combine"):
        return int(lib.tps_crc32_combine(crc1 & 0xFFFFFFFF, crc2 & 0xFFFFFFFF, int(len2)))
    if len2 <= 0:
        return crc1

    def times(mat, vec):
        s, i = 0, 0
        while vec:
            if vec & 1:
                s ^= mat[i]
            vec >>= 1
            i += 1
        return s

    def square(mat):
        return [times(mat, mat[n]) for n in range(32)]
    odd = [0xEDB88320] + [1 << n for n in range(31)]
    even = square(odd)
    odd = square(even)
    while True:
        even = square(odd)
        if len2 & 1:
            crc1 = times(even, crc1)
        len2 >>= 1
        if not len2:
            break
        odd = square(even)
        if len2 & 1:
            crc1 = times(odd, crc1)
        len2 >>= 1
        if not len2:
            break
    return crc1 ^ crc2


def _local_header(name: bytes, crc: int, size: int) -> bytes:
    extra = struct.pack("<HHQQ", 1, 16, size, size)
    return struct.pack("<IHHHHHIIIHH", 0x04034B50, 45, 0, 0, _DOS_TIME, _DOS_DATE, crc & 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF,
                       len(name), len(extra)) + name + extra


def _central_entry(name: bytes, crc: int, size: int, offset: int) -> bytes:
    extra = struct.pack("<HHQQQ", 1, 24, size, size, offset)
    return struct.pack("<IHHHHHHIIIHHHHHII", 0x02014B50, (3 << 8) | 45, 45, 0, 0, _DOS_TIME, _DOS_DATE, crc & 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF,
                       len(name), len(extra), 0, 0, 0, 0o100644 << 16, 0xFFFFFFFF) + name + extra


def _end_records(n_entries: int, cd_size: int, cd_offset: int) -> bytes:
    z64 = struct.pack("<IQHHIIQQQQ", 0x06064B50, 44, 45, 45, 0, 0, n_entries, n_entries, cd_size, cd_offset)
    loc = struct.pack("<IIQI", 0x07064B50, 0, cd_offset + cd_size, 1)
    end = struct.pack("<IHHHHIIH", 0x06054B50, 0, 0, min(n_entries, 0xFFFF), min(n_entries, 0xFFFF), 0xFFFFFFFF, 0xFFFFFFFF, 0)
    return z64 + loc + end


def _npy_bytes(arr) -> bytes:
    import io
    h = io.BytesIO()
    np.lib.format.write_array(h, np.asanyarray(arr), allow_pickle=False)
    return h.getvalue()


def _counts_npy_header(rows: int, p: int) -> bytes:
    """A version-1.0 .npy header of exactly DATA_OFF - _LOCAL_LEN bytes for a u8[rows, p] array (blank-padded: numpy pads its own
    headers the same way, to 64 bytes)."""
    total = DATA_OFF - _LOCAL_LEN
    text = "{'descr': '|u1', 'fortran_order': False, 'shape': (%d, %d), }" % (rows, p)
    hlen = total - 10
    body = text.encode("latin1")
    assert len(body) < hlen
    return b"\x93NUMPY\x01\x00" + struct.pack("<H", hlen) + body + b" " * (hlen - len(body) - 1) + b"\n"


class RawNpzWriter:
    """One archive (one input file, one k).  Thread roles: the batch workers call `write_batch` (any order, each batch index once),
    the consumer calls `add_reads` in file order, then `finish` (or `discard`)."""

    def __init__(self, path, pattern, slide, read_check=None):
        self.path = path
        self.pattern, self.slide = list(pattern), int(slide)
        self.read_check = read_check
        self.read_id, self.tail, self.n_win = [], [], []
        self.fd = None
        self._cv = threading.Condition()
        self._next_seq = 0
        self._end = 0                       # bytes of rows claimed so far
        self._blocks = {}                   # batch index -> (bytes, crc)
        self._aborted = False

    # ---- workers
    def select(self, pb, res):
        """The reads of a batch whose rows are kept: those that pass, or of those the one `--read_check` names."""
        keep = np.nonzero(res["pass"])[0]
        if self.read_check is not None and len(keep):
            keep = keep[[pb.read_id(int(i)) == self.read_check for i in keep]]
        return keep

    def _claim(self, seq, nbytes):
        with self._cv:
            while self._next_seq != seq:
                if self._aborted:
                    raise RuntimeError("raw-count writer aborted")
                self._cv.wait(0.2)
            off = self._end
            self._end += int(nbytes)
            self._next_seq += 1
            if self.fd is None and nbytes:
                self.fd = os.open(self.path, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644)
            self._cv.notify_all()
            return off

    def write_batch(self, seq, eng, slot, slot_reads, n_rows):
        """Rows of the reads `slot_reads` (ascending indices into the engine's resident batch in `slot`), `n_rows` windows in all."""
        nbytes = int(n_rows) * len(self.pattern)
        off = self._claim(seq, nbytes)
        crc = 0
        if nbytes:
            got, crc = eng.raw_to_fd(slot, np.ascontiguousarray(slot_reads, np.int64), self.fd, DATA_OFF + off)
            if got != nbytes:
                raise RuntimeError(f"raw rows: {got} bytes written, {nbytes} expected")
        with self._cv:
            self._blocks[seq] = (nbytes, crc)

    def skip_batch(self, seq):
        """A batch that contributes no rows still takes its turn."""
        self._claim(seq, 0)
        with self._cv:
            self._blocks[seq] = (0, 0)

    def abort(self):
        with self._cv:
            self._aborted = True
            self._cv.notify_all()

    # ---- consumer
    def add_reads(self, ids, tails, n_win):
        self.read_id += list(ids)
        self.tail += list(tails)
        self.n_win += [int(x) for x in n_win]

    def discard(self):
        if self.fd is not None:
            os.close(self.fd)
            self.fd = None
            try:
                os.unlink(self.path)
            except OSError:
                pass

    def finish(self):
        if not self.read_id:
            self.discard()
            return
        p = len(self.pattern)
        n_win = np.array(self.n_win, dtype=np.int64)
        win_off = np.zeros(len(n_win) + 1, dtype=np.int64)
        np.cumsum(n_win, out=win_off[1:])
        rows = int(win_off[-1])
        if rows * p != self._end:
            self.discard()
            raise RuntimeError(f"raw rows: {self._end} bytes written for {rows} windows of {p} patterns")
        if self.fd is None:                                    # passing reads, none with a window
            self.fd = os.open(self.path, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644)
        fd = self.fd
        head = _counts_npy_header(rows, p)
        crc, total = zlib.crc32(head), len(head)
        for seq in sorted(self._blocks):
            nb, c = self._blocks[seq]
            if nb:
                crc = _crc32_combine(crc, c, nb)
                total += nb
        os.pwrite(fd, _local_header(_NAME, crc, total) + head, 0)
        entries = [_central_entry(_NAME, crc, total, 0)]
        pos = DATA_OFF + self._end
        small = dict(read_id=np.array(self.read_id), tail=np.array(self.tail), win_off=win_off, pattern=np.array(self.pattern),
                     slide=np.int64(self.slide))
        for name, arr in small.items():
            blob = _npy_bytes(arr)
            nm = (name + ".npy").encode()
            c = zlib.crc32(blob)
            rec = _local_header(nm, c, len(blob)) + blob
            os.pwrite(fd, rec, pos)
            entries.append(_central_entry(nm, c, len(blob), pos))
            pos += len(rec)
        cd = b"".join(entries)
        os.pwrite(fd, cd + _end_records(len(entries), len(cd), pos), pos)
        os.ftruncate(fd, pos + len(cd) + 98)
        os.close(fd)
        self.fd = None
