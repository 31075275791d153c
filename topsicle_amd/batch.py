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
    """Fused scan of one batch of records.  Returns (results, sums, raw, win_off)."""
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
