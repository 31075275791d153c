"""Batched driver: parse a file once, cut the record stream into batches balanced by bases,
run the fused scan (one launch per batch) and hand back per-read rows.

Replaces the O(R_pass^2) structure of the reference's process_file loop, which re-parses the
filtered file once per passing read (main.py:125-152 -> allsteps.py:257-259).
"""
from __future__ import annotations

import numpy as np

from . import hiplib

BATCH_BASES = 256 << 20          # ~256 MB of bases per upload
BATCH_READS = 1 << 20


def record_batches(records, max_bases: int = BATCH_BASES, max_reads: int = BATCH_READS):
    """Group an iterator of records into lists holding at most max_bases bases."""
    cur, nb = [], 0
    for rec in records:
        if cur and (nb + len(rec.seq) > max_bases or len(cur) >= max_reads):
            yield cur
            cur, nb = [], 0
        cur.append(rec)
        nb += len(rec.seq)
    if cur:
        yield cur


def scan_records(engine, recs, prm: hiplib.Params, slot: int = 0, want_sums=False, want_raw=False):
    """Fused scan of one batch (a list of records or a seqio.RecordBatch).  Returns (results, sums, raw, win_off)."""
    if hasattr(recs, "bases"):
        bases, offsets = recs.bases, recs.offsets
    else:
        bases, offsets = hiplib.pack_reads([r.seq for r in recs])
    engine.upload(slot, bases, offsets)
    flags = prm.flags | (hiplib.F_STORE_SUMS if want_sums else 0) | (hiplib.F_STORE_RAW if want_raw else 0)
    p = hiplib.Params.from_buffer_copy(prm)
    p.flags = flags
    engine.scan(slot, p)
    engine.sync()
    res = engine.results(slot)
    sums = raw = win_off = None
    if want_sums:
        sums, win_off = engine.window_sums(slot)
    if want_raw:
        raw, win_off = engine.window_raw(slot)
    return res, sums, raw, win_off


class EnginePool:
    """Read-sharding over the GPUs of one node: one engine (context) per
    GPU, each driven by its own host thread; batches of ~equal bases are dealt round-robin and
    results are handed back in input order.  No collective, no device-to-device traffic."""

    def __init__(self, engines, patterns):
        self.engines = list(engines)
        if not self.engines:
            raise ValueError("no engines")
        for e in self.engines:
            e.set_patterns(patterns)

    def scan_file(self, filepath, prm, want_sums=False, want_raw=False, max_bases=None):
        """Like scan_stream over the records of a FASTA/FASTQ(.gz) file, decoded natively into batch
        buffers (seqio.read_batches); yields (RecordBatch, results, sums, raw, win_off)."""
        from . import seqio
        n = len(self.engines)
        if max_bases is None:
            max_bases = BATCH_BASES if n == 1 else BATCH_BASES // 4
        return self._scan_batches(seqio.read_batches(filepath, max_bases=max_bases), prm, want_sums, want_raw)

    def scan_stream(self, records, prm, want_sums=False, want_raw=False, max_bases=None):
        n = len(self.engines)
        if max_bases is None:
            max_bases = BATCH_BASES if n == 1 else BATCH_BASES // 4
        return self._scan_batches(record_batches(records, max_bases=max_bases), prm, want_sums, want_raw)

    def _scan_batches(self, batches, prm, want_sums, want_raw):
        # one worker thread per engine, two batches in flight each: the host decodes / packs batch i+1 (native code, the
        # GIL is released) while the GPU scans batch i and the caller writes the results of batch i-1 -- also with ONE GPU
        n = len(self.engines)
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        workers = [ThreadPoolExecutor(max_workers=1) for _ in range(n)]
        pending = deque()
        try:
            for i, recs in enumerate(batches):
                eng = self.engines[i % n]
                fut = workers[i % n].submit(scan_records, eng, recs, prm, 0, want_sums, want_raw)
                pending.append((recs, fut))
                while len(pending) >= 2 * n:
                    r, f = pending.popleft()
                    yield (r,) + f.result()
            while pending:
                r, f = pending.popleft()
                yield (r,) + f.result()
        finally:
            for w in workers:
                w.shutdown(wait=True)
