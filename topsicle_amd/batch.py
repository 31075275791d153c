"""Batched driver: parse a file once, cut the record stream into batches balanced by bases, pack them (2 bits per base),
run the fused scan (one launch per batch) and hand back per-read rows.

Replaces the O(R_pass^2) structure of the reference's process_file loop, which re-parses the filtered file once per
passing read (main.py:125-152 -> allsteps.py:257-259), and its `Pool` over input files (main.py:232-235) as the unit of
parallelism: here a reader thread decodes + packs batches into pinned staging buffers while worker threads -- one per
context, two contexts per GPU -- pull batches from ONE shared queue (dynamic balancing over all GPUs, SURVEY.md section
8e), upload, scan and download; the caller consumes results in input order.  No collective, no device-to-device traffic.
"""
from __future__ import annotations

import os
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


class Job:
    """One scan of a resident batch: a pattern table + parameters + which window outputs to download.  Several jobs per
    batch = `--telophrase 4 5 6`: the batch is parsed, packed and uploaded ONCE and scanned once per k (the reference
    re-parses the input for every k, main.py:206-235)."""

    def __init__(self, patterns, prm: hiplib.Params, want_sums=False, want_raw=False):
        self.patterns, self.prm, self.want_sums, self.want_raw = list(patterns), prm, bool(want_sums), bool(want_raw)


def upload_batch(engine, recs, slot: int = 0):
    """A seqio.PackedBatch is uploaded as it is (3 bits per base); a seqio.RecordBatch or a list of records goes up as
    ASCII and is packed on the device."""
    if hasattr(recs, "seq2"):
        engine.upload_packed(slot, recs.seq2, recs.inv if recs.any_invalid else None, recs.desc)
    elif hasattr(recs, "bases"):
        engine.upload(slot, recs.bases, recs.offsets)
    else:
        engine.upload(slot, *hiplib.pack_reads([r.seq for r in recs]))


def scan_jobs(engine, recs, jobs, slot: int = 0):
    """Upload once, then one fused scan per job.  Returns [(results, sums, raw, win_off), ...] in job order.

    Several jobs (pattern tables) on a HipScanner run AT THE SAME TIME: job j > 0 goes to the engine's j-th helper context,
    which borrows the resident batch (tps_batch_share) and keeps its own table -- no table switch per batch, and the launches
    of the k passes overlap on the GPU (measured: 603 vs 785 us per 10 000 x 25 kb batch for k = 4, 5, 6).
    TOPSICLE_SEQUENTIAL_TABLES=1 scans them back to back on the one context instead (the A/B switch)."""
    out = []
    concurrent = len(jobs) > 1 and hasattr(engine, "helper") and os.environ.get("TOPSICLE_SEQUENTIAL_TABLES", "0") != "1"
    engines = [engine] + ([engine.helper(j) for j in range(len(jobs) - 1)] if concurrent else [engine] * (len(jobs) - 1))
    try:
        upload_batch(engine, recs, slot)
        prms = []
        for job in jobs:
            p = hiplib.Params.from_buffer_copy(job.prm)
            p.flags = job.prm.flags | (hiplib.F_STORE_SUMS if job.want_sums else 0) | (hiplib.F_STORE_RAW if job.want_raw else 0)
            prms.append(p)
        if concurrent:
            for eng, job, p in zip(engines, jobs, prms):
                if eng is not engine:
                    eng.share(slot, engine, slot)
                if getattr(eng, "patterns", None) != job.patterns:
                    eng.set_patterns(job.patterns)
                eng.scan(slot, p)
        for n, (eng, job, p) in enumerate(zip(engines, jobs, prms)):
            if not concurrent:
                if len(jobs) > 1 or getattr(eng, "patterns", None) != job.patterns:
                    eng.set_patterns(job.patterns)
                eng.scan(slot, p)
            eng.sync()
            if n == 0 and hasattr(recs, "release"):
                recs.release()             # the upload has completed: the staging buffers go back to the reader
            res = eng.results(slot)
            if p.flags & hiplib.F_BINSEG:              # exact ties: ruptures' float64 answer (a handful of reads at most)
                hiplib.resolve_ties(eng, slot, res, len(job.patterns), p.jump, p.min_size)
            sums = raw = win_off = None
            if job.want_sums:
                sums, win_off = eng.window_sums(slot)
            if job.want_raw:
                raw, win_off = eng.window_raw(slot)
            out.append((res, sums, raw, win_off))
    except BaseException:
        for eng in dict.fromkeys(engines):
            try:
                eng.sync()                 # an asynchronous upload / a borrowed batch may still be in use
            except Exception:
                pass
        if hasattr(recs, "release"):
            recs.release()                 # a failed batch must not take its staging buffers with it
        raise
    return out


def scan_records(engine, recs, prm: hiplib.Params, slot: int = 0, want_sums=False, want_raw=False):
    """Fused scan of one batch with the engine's current pattern table.  Returns (results, sums, raw, win_off)."""
    return scan_jobs(engine, recs, [Job(engine.patterns, prm, want_sums, want_raw)], slot)[0]


class EnginePool:
    """Read sharding over the contexts of one node: `engines` holds one or more contexts per GPU (two per GPU overlap the
    upload / launch ramp / result download of one batch with the scan of another).  Batches go to whichever context is
    free next; results come back in input order."""

    def __init__(self, engines, patterns=None):
        self.engines = list(engines)
        if not self.engines:
            raise ValueError("no engines")
        self.patterns = None if patterns is None else list(patterns)
        if self.patterns is not None:
            for e in self.engines:
                e.set_patterns(self.patterns)

    # -- sources
    def scan_file(self, filepath, prm, want_sums=False, want_raw=False, max_bases=None):
        """Records of a FASTA/FASTQ(.gz) file, decoded and packed natively (seqio.read_batches_packed); yields
        (PackedBatch, results, sums, raw, win_off) in file order."""
        from . import seqio
        max_bases = max_bases or BATCH_BASES
        words_cap = max(max_bases // 16, 1024)                 # (a read's padding to whole 64-base quads counts too)
        reads_cap = min(BATCH_READS, max(64, max_bases // 256))
        pool = self._staging_pool(words_cap, reads_cap)
        return self._single(self._run(seqio.read_batches_packed(filepath, pool, max_records=reads_cap),
                                      [Job(self.patterns, prm, want_sums, want_raw)], pool))

    def scan_file_jobs(self, filepath, jobs, max_bases=None):
        """One pass over the file for several jobs (pattern tables): yields (PackedBatch, [(results, sums, raw, win_off) per job])."""
        from . import seqio
        max_bases = max_bases or BATCH_BASES
        words_cap = max(max_bases // 16, 1024)
        reads_cap = min(BATCH_READS, max(64, max_bases // 256))
        pool = self._staging_pool(words_cap, reads_cap)
        return self._run(seqio.read_batches_packed(filepath, pool, max_records=reads_cap), list(jobs), pool)

    def _staging_pool(self, words_cap, reads_cap):
        """The pinned staging buffers of this engine set: allocated ONCE per (engine set, geometry) and kept on the first engine,
        so that a new EnginePool per input file (main.process_file_multi) reuses them -- they are only freed with the context
        (a pool per file leaked ~28 MB of pinned memory per buffer set and file: ADVICE r2).  Every context uploads from them
        asynchronously (the library keeps a process-wide registry of pinned buffers)."""
        from . import seqio
        owner = self.engines[0]
        pools = getattr(owner, "_staging_pools", None)
        if pools is None:
            pools = owner._staging_pools = {}
        key = (words_cap, reads_cap, len(self.engines) + 2)
        if key not in pools:
            pools[key] = seqio.BufferPool(len(self.engines) + 2, words_cap, reads_cap, getattr(owner, "host_alloc", None))
        return pools[key]

    def scan_stream(self, records, prm, want_sums=False, want_raw=False, max_bases=None):
        return self._single(self._run(record_batches(records, max_bases=max_bases or BATCH_BASES), [Job(self.patterns, prm, want_sums, want_raw)]))

    @staticmethod
    def _single(it):
        for b, outs in it:
            yield (b,) + outs[0]

    # -- the pipeline
    def _run(self, batches, jobs, pool=None):
        n = len(self.engines)
        if pool is not None:
            pool.abort.clear()
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
                        if hasattr(b, "release"):
                            b.release()
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
                    q_out.put(("batch", i, (b, scan_jobs(eng, b, jobs, 0))))
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
            if pool is not None:
                pool.abort.set()                       # a reader waiting for a staging buffer gives up
            try:
                while True:
                    item = q_in.get_nowait()
                    if item is not None and hasattr(item[1], "release"):
                        item[1].release()              # batches nobody will scan hand their staging buffers back
            except queue.Empty:
                pass
            for _ in range(n):
                try:
                    q_in.put_nowait(None)
                except queue.Full:
                    pass
            for t in threads:
                t.join(timeout=5)
