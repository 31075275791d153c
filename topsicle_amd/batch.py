"""Batched driver: parse a file once, cut the record stream into batches balanced by bases, pack them (2 bits per base),
run the fused scan (one launch per batch) and hand back per-read rows.

Replaces the O(R_pass^2) structure of the reference's process_file loop, which re-parses the filtered file once per
passing read (main.py:125-152 -> allsteps.py:257-259), and its `Pool` over input files (main.py:232-235) as the unit of
parallelism: here a reader thread decodes + packs batches into pinned staging buffers while worker threads -- one per
context, two contexts per GPU -- pull batches from ONE shared queue (dynamic balancing over all GPUs, SURVEY.md section
8e), upload, scan and download; the caller consumes results in input order.  No collective, no device-to-device traffic.
"""
from __future__ import annotations

import queue
import threading

import numpy as np

from . import hiplib

BATCH_BASES = 64 << 20           # bases per upload: 16 MB of packed words + 8 MB of invalid masks across PCIe
BATCH_READS = 1 << 18


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
    """Fused scan of one batch: a seqio.PackedBatch (uploaded as it is), a seqio.RecordBatch or a list of records (ASCII
    upload, packed on the device).  Returns (results, sums, raw, win_off)."""
    if hasattr(recs, "seq2"):
        engine.upload_packed(slot, recs.seq2, recs.inv if recs.any_invalid else None, recs.desc)
    elif hasattr(recs, "bases"):
        engine.upload(slot, recs.bases, recs.offsets)
    else:
        engine.upload(slot, *hiplib.pack_reads([r.seq for r in recs]))
    flags = prm.flags | (hiplib.F_STORE_SUMS if want_sums else 0) | (hiplib.F_STORE_RAW if want_raw else 0)
    p = hiplib.Params.from_buffer_copy(prm)
    p.flags = flags
    engine.scan(slot, p)
    engine.sync()
    if hasattr(recs, "release"):
        recs.release()                 # the staging buffers are free again: the reader may refill them
    res = engine.results(slot)
    sums = raw = win_off = None
    if want_sums:
        sums, win_off = engine.window_sums(slot)
    if want_raw:
        raw, win_off = engine.window_raw(slot)
    return res, sums, raw, win_off


class EnginePool:
    """Read sharding over the contexts of one node: `engines` holds one or more contexts per GPU (two per GPU overlap the
    upload / launch ramp / result download of one batch with the scan of another).  Batches go to whichever context is
    free next; results come back in input order."""

    def __init__(self, engines, patterns):
        self.engines = list(engines)
        if not self.engines:
            raise ValueError("no engines")
        for e in self.engines:
            e.set_patterns(patterns)

    # -- sources
    def scan_file(self, filepath, prm, want_sums=False, want_raw=False, max_bases=None):
        """Records of a FASTA/FASTQ(.gz) file, decoded and packed natively (seqio.read_batches_packed); yields
        (PackedBatch, results, sums, raw, win_off) in file order."""
        from . import seqio
        max_bases = max_bases or BATCH_BASES
        words_cap = max(max_bases // 16, 1024)                 # (a read's padding to whole 64-base quads counts too)
        reads_cap = min(BATCH_READS, max(64, max_bases // 256))
        key = (words_cap, reads_cap)
        if getattr(self, "_pool_key", None) != key:            # pinned staging buffers are allocated once and reused file after file
            alloc = getattr(self.engines[0], "host_alloc", None)
            self._pool = seqio.BufferPool(len(self.engines) + 2, words_cap, reads_cap, alloc)
            self._pool_key = key
        return self._run(seqio.read_batches_packed(filepath, self._pool, max_records=reads_cap), prm, want_sums, want_raw)

    def scan_stream(self, records, prm, want_sums=False, want_raw=False, max_bases=None):
        return self._run(record_batches(records, max_bases=max_bases or BATCH_BASES), prm, want_sums, want_raw)

    # -- the pipeline
    def _run(self, batches, prm, want_sums, want_raw):
        n = len(self.engines)
        q_in: queue.Queue = queue.Queue(maxsize=n + 1)
        q_out: queue.Queue = queue.Queue()
        stop = threading.Event()

        def reader():
            count = 0
            try:
                for b in batches:
                    while not stop.is_set():
                        try:
                            q_in.put((count, b), timeout=0.2)
                            break
                        except queue.Full:
                            continue
                    if stop.is_set():
                        break
                    count += 1
                q_out.put(("eof", count, None))
            except BaseException as e:          # parse errors surface in the consumer
                q_out.put(("error", None, e))
            finally:
                for _ in range(n):
                    q_in.put(None)

        def worker(eng):
            try:
                while True:
                    item = q_in.get()
                    if item is None:
                        return
                    i, b = item
                    q_out.put(("batch", i, (b,) + scan_records(eng, b, prm, 0, want_sums, want_raw)))
            except BaseException as e:
                stop.set()
                q_out.put(("error", None, e))

        threads = [threading.Thread(target=reader, daemon=True)] + [threading.Thread(target=worker, args=(e,), daemon=True) for e in self.engines]
        for t in threads:
            t.start()
        pending, nxt, total = {}, 0, None
        try:
            while total is None or nxt < total:
                kind, i, payload = q_out.get()
                if kind == "error":
                    raise payload
                if kind == "eof":
                    total = i
                    continue
                pending[i] = payload
                while nxt in pending:
                    yield pending.pop(nxt)
                    nxt += 1
        finally:
            stop.set()
            try:
                while True:
                    q_in.get_nowait()
            except queue.Empty:
                pass
            for _ in range(n):
                try:
                    q_in.put_nowait(None)
                except queue.Full:
                    pass
            for t in threads:
                t.join(timeout=5)
