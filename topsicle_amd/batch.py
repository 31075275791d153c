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
SEQUENTIAL_TABLES = False        # True: several pattern tables of one batch are scanned back to back on the one context instead of
                                 # at the same time on helper contexts (the A/B switch of tests and bench.py --sequential-tables)


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

    def __init__(self, patterns, prm: hiplib.Params, want_sums=False, want_raw=False, raw_sink=None):
        """raw_sink (with want_raw; rawnpz.RawNpzWriter): the raw rows of the reads the sink selects are written to its file by
        the worker that scanned the batch, device -> file, and the caller gets a `RawKept` in place of the rows."""
        self.patterns, self.prm, self.want_sums, self.want_raw = list(patterns), prm, bool(want_sums), bool(want_raw)
        self.raw_sink = raw_sink if want_raw else None


class RawKept:
    """What a job with a raw_sink returns in place of the raw rows: `keep` = the reads (indices into the batch) whose rows the
    sink has written, in that order."""

    def __init__(self, keep):
        self.keep = keep


def _sink_rows(job, seq, eng, slot, recs, res, slot_index=None):
    """Hand the rows of one scanned batch to the job's sink.  `slot_index`: the batch indices of the reads resident in `slot`
    when that is not the whole batch (second pass of the two-pass route: the passing reads only)."""
    sink = job.raw_sink
    keep = sink.select(recs, res)
    if not len(keep):
        sink.skip_batch(seq)
        return RawKept(keep)
    slot_reads = keep if slot_index is None else np.searchsorted(slot_index, keep)
    sink.write_batch(seq, eng, slot, slot_reads, int(res["n_win"][keep].astype(np.int64).sum()))
    return RawKept(keep)


def upload_batch(engine, recs, slot: int = 0):
    """A seqio.PackedBatch is uploaded as it is (3 bits per base); a seqio.RecordBatch or a list of records goes up as
    ASCII and is packed on the device."""
    if hasattr(recs, "seq2"):
        recs.uploaded_bytes = 4 * len(recs.seq2) + (2 * len(recs.inv) if recs.any_invalid else 0) + 16 * len(recs.desc)
        engine.upload_packed(slot, recs.seq2, recs.inv if recs.any_invalid else None, recs.desc)
    elif hasattr(recs, "bases"):
        engine.upload(slot, recs.bases, recs.offsets)
    else:
        engine.upload(slot, *hiplib.pack_reads([r.seq for r in recs]))


TWO_PASS_MAX_PASSING = 0.25      # heads mode stays on while at most this share of a batch's reads passes step 1
PROBE_READS = 64                 # auto mode: reads of the probe batch that opens a file
AUTO_MIN_FILE_BYTES = 1 << 30    # auto mode only probes files of at least this size (a quarter of it for .gz): the probe costs ~0.5 ms per
                                 # file, and what two passes save on a small file is less than that


def scan_jobs_heads(engine, recs, jobs, slot: int = 0, seq=None):
    """Two passes over a batch that came in HEADS mode (seqio.PackedBatch.full_len: only the first + last no_bp bases of every read
    were packed and uploaded -- all that step 1, allsteps.py:174-198, looks at):
      A  step 1 on the heads, one launch per job (table);
      B  the reads that pass -- in real WGS data well under 1 % (the reference's README: "> 20 GB and / or > 1 million reads") -- get
         the part of them step 2 scans, their first / last min(L, maxlengthtelo) bases (allsteps.py:263-271), packed from the batch's
         text (seqio.pack_spans) and uploaded as a small batch of its own with the tails step 1 chose; windows + change point there.
    The rows that come back are those of the one-pass scan (tests/test_two_pass.py: equal field by field); sums / raw rows are
    laid out over ALL reads of the batch with no windows for the reads that do not pass.  PCIe bytes per input base, 30 kb reads of
    which 1 % pass: 0.375 -> 0.028.  Same return value as scan_jobs."""
    import numpy as np
    concurrent = len(jobs) > 1 and hasattr(engine, "helper") and not SEQUENTIAL_TABLES
    engines = [engine] + ([engine.helper(j) for j in range(len(jobs) - 1)] if concurrent else [engine] * (len(jobs) - 1))
    slot_b = slot + 1
    full = np.asarray(recs.full_len, np.int64)
    out = []
    try:
        upload_batch(engine, recs, slot)
        prm_a = []
        for job in jobs:
            p = hiplib.Params.from_buffer_copy(job.prm)
            p.flags = hiplib.F_STEP1
            p.min_len = 0                              # (the pseudo-reads are short: the length filter is applied to full_len below)
            prm_a.append(p)
        res_a = []
        if concurrent:
            for eng, job, p in zip(engines, jobs, prm_a):
                if eng is not engine:
                    eng.share(slot, engine, slot)
                if getattr(eng, "patterns", None) != job.patterns:
                    eng.set_patterns(job.patterns)
                eng.scan(slot, p)
        for n, (eng, job, p) in enumerate(zip(engines, jobs, prm_a)):
            if not concurrent:
                if len(jobs) > 1 or getattr(eng, "patterns", None) != job.patterns:
                    eng.set_patterns(job.patterns)
                eng.scan(slot, p)
            eng.sync()
            res_a.append(eng.results(slot))
        recs.release()                                 # the heads are on the device: the staging buffers go back to the reader
        uploaded, holds = {}, {}                       # (passing reads, tails, maxlen) -> the engine that uploaded them; engine -> what its slot_b holds
        for n, (eng, job, ra) in enumerate(zip(engines, jobs, res_a)):
            res = ra.copy()
            passing = (ra["pass"] != 0) & (full > job.prm.min_len)
            res["pass"] = passing
            res["n_win"] = 0
            res["bkp"] = -1
            res["gain"] = 0.0
            res["flags"] = 0
            idx = np.nonzero(passing)[0]
            sums = raw = None
            win_off = np.zeros(len(res) + 1, np.int64)
            want_b = (job.prm.flags & (hiplib.F_WINDOWS | hiplib.F_BINSEG)) != 0
            if len(idx) and want_b:
                tails = ra["tail"][idx].astype(np.uint8)
                key = (idx.tobytes(), tails.tobytes(), int(job.prm.maxlen))
                if not concurrent and len(jobs) > 1 and getattr(eng, "patterns", None) != job.patterns:
                    eng.set_patterns(job.patterns)
                if holds.get(id(eng)) != key:
                    owner = uploaded.get(key)
                    if owner is not None and owner is not eng and holds.get(id(owner)) == key and hasattr(eng, "share"):
                        eng.share(slot_b, owner, slot_b)           # another table's context holds exactly these reads already
                    else:
                        seq2, inv, desc = _pack_passing(recs, idx, tails, int(job.prm.maxlen))
                        eng.upload_packed(slot_b, seq2, inv if (desc["flags"] & 1).any() else None, desc)
                        recs.uploaded_bytes += 4 * len(seq2) + (2 * len(inv) if (desc["flags"] & 1).any() else 0) + 16 * len(desc)
                        uploaded[key] = eng
                    holds[id(eng)] = key
                eng.set_tails(slot_b, tails)
                p = hiplib.Params.from_buffer_copy(job.prm)
                p.flags = (job.prm.flags & ~hiplib.F_STEP1) | hiplib.F_TAILS_IN | (hiplib.F_STORE_SUMS if job.want_sums else 0) | \
                          (hiplib.F_STORE_RAW if job.want_raw else 0)
                p.min_len = 0
                eng.scan(slot_b, p)
                eng.sync()
                rb = eng.results(slot_b)
                if p.flags & hiplib.F_BINSEG:
                    hiplib.resolve_ties(eng, slot_b, rb, len(job.patterns), p.jump, p.min_size)
                for f in ("n_win", "bkp", "gain", "flags"):
                    res[f][idx] = rb[f]
                np.cumsum(np.where(passing, res["n_win"], 0), out=win_off[1:])
                if job.want_sums:
                    sums, wo_b = eng.window_sums(slot_b)
                    assert int(wo_b[-1]) == int(win_off[-1])
                if job.want_raw and job.raw_sink is not None and seq is not None:
                    raw = _sink_rows(job, seq, eng, slot_b, recs, res, slot_index=idx)
                elif job.want_raw:
                    raw, wo_b = eng.window_raw(slot_b)
                    assert int(wo_b[-1]) == int(win_off[-1])
            else:
                if job.want_sums:
                    sums = np.zeros(0, np.int32)
                if job.want_raw and job.raw_sink is not None and seq is not None:
                    job.raw_sink.skip_batch(seq)
                    raw = RawKept(np.zeros(0, np.int64))
                elif job.want_raw:
                    raw = np.zeros((0, len(job.patterns)), np.uint8)
            out.append((res, sums, raw, win_off if (job.want_sums or job.want_raw) else None))
    except BaseException:
        for eng in dict.fromkeys(engines):
            try:
                eng.sync()
            except Exception:
                pass
        recs.release()
        raise
    return out


def _pack_passing(recs, idx, tails, maxlen):
    """The scanned part of the passing reads as a packed batch of its own: from the text the batch's spans point into, or -- a
    batch that has none (ASCII batches of odd inputs never come in heads mode) -- an error."""
    from . import seqio
    if recs.spans is None or recs.text is None:
        raise RuntimeError("heads-mode batch without text spans")
    return seqio.pack_spans(recs, idx, tails, maxlen)


def scan_jobs(engine, recs, jobs, slot: int = 0, seq=None):
    """Upload once, then one fused scan per job.  Returns [(results, sums, raw, win_off), ...] in job order.
    `seq`: the batch's index in its file (EnginePool passes it): jobs with a raw_sink then write their rows through it.
    A batch in heads mode (PackedBatch.full_len) takes the two-pass route: scan_jobs_heads.

    Several jobs (pattern tables) on a HipScanner run AT THE SAME TIME: job j > 0 goes to the engine's j-th helper context,
    which borrows the resident batch (tps_batch_share) and keeps its own table -- no table switch per batch, and the launches
    of the k passes overlap on the GPU (measured: 603 vs 785 us per 10 000 x 25 kb batch for k = 4, 5, 6).
    batch.SEQUENTIAL_TABLES scans them back to back on the one context instead (the A/B switch)."""
    if getattr(recs, "full_len", None) is not None:
        return scan_jobs_heads(engine, recs, jobs, slot, seq)
    out = []
    concurrent = len(jobs) > 1 and hasattr(engine, "helper") and not SEQUENTIAL_TABLES
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
            if job.want_raw and job.raw_sink is not None and seq is not None:
                win_off = eng.window_offsets(slot)
                raw = _sink_rows(job, seq, eng, slot, recs, res)
            elif job.want_raw:
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

    def __init__(self, engines, patterns=None, two_pass=None):
        """two_pass: "auto" (heads mode while few reads pass step 1: see _heads_mode), "on", "off"; None = $TOPSICLE_TWO_PASS or auto."""
        self.engines = list(engines)
        # TOPSICLE_TWO_PASS is only the default of callers that say nothing (it used to override an explicit argument)
        if two_pass is None:
            two_pass = os.environ.get("TOPSICLE_TWO_PASS", "auto")
        if two_pass not in ("auto", "on", "off"):
            raise ValueError(f"two_pass must be auto, on or off, not {two_pass!r}")
        self.two_pass = two_pass
        self.stats = {"batches": 0, "heads_batches": 0, "upload_bytes": 0, "input_bases": 0}
        self._stats_lock = threading.Lock()           # (_heads_feedback runs on the worker threads)
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
        jobs = [Job(self.patterns, prm, want_sums, want_raw)]
        hb, hm = self._heads_mode(jobs, filepath)
        return self._single(self._run(seqio.read_batches_packed(filepath, pool, max_records=reads_cap, heads_bp=hb,
                                                                first_batch_records=self._probe_records(hm)), jobs, pool, hm))

    def scan_file_jobs(self, filepath, jobs, max_bases=None, shards=1, shard_min_bytes=64 << 20):
        """One pass over the file for several jobs (pattern tables): yields (PackedBatch, [(results, sums, raw, win_off) per job]).
        shards > 1 (round 5): ONE plain file is cut into that many byte ranges, each decoded by a reader thread team of its own at the
        same time (seqio.shard_ranges / tps_reader_open_range), all of them feeding this pool's contexts; batches come back in FILE
        order.  The reference tells users to split a file of "> 20 GB and / or > 1 million reads" by hand and run the pieces
        (README.md:267-268; its unit of parallelism is the file: main.py:232-235).  Only jobs whose results are per-read records
        (no window sums / raw rows handed back, no raw sink: those are laid out in batch order); a BGZF file is cut at block boundaries,
        an ordinary .gz (one deflate stream) keeps one reader."""
        from . import seqio
        max_bases = max_bases or BATCH_BASES
        words_cap = max(max_bases // 16, 1024)
        reads_cap = min(BATCH_READS, max(64, max_bases // 256))
        jobs = list(jobs)
        ranges = None
        if shards and shards > 1 and not any(j.want_sums or j.want_raw for j in jobs):
            ranges = seqio.shard_ranges(filepath, shards, shard_min_bytes)
        if not ranges:
            pool = self._staging_pool(words_cap, reads_cap)
            hb, hm = self._heads_mode(jobs, filepath)
            return self._run(seqio.read_batches_packed(filepath, pool, max_records=reads_cap, heads_bp=hb, first_batch_records=self._probe_records(hm)),
                             jobs, pool, hm)
        return self._scan_sharded(filepath, jobs, ranges, words_cap, reads_cap)

    def _scan_sharded(self, filepath, jobs, ranges, words_cap, reads_cap):
        from . import seqio
        n = len(ranges)
        # every shard's reader gets its share of the host's cores (the teams run at the same time) and staging buffers of its own
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        threads = max(1, min(32, cores) // n) if cores >= 2 * n else 1
        pool = self._staging_pool(words_cap, reads_cap, sets=len(self.engines) + 2 * n)
        hb, hm = self._heads_mode(jobs, filepath)
        if hm["auto"]:
            hm["first"].set()                          # (no probe batch per shard: auto mode starts from whole reads and switches on the first verdict)
        infos = [dict() for _ in ranges]
        sources = [seqio.read_batches_packed(filepath, pool, max_records=reads_cap, heads_bp=hb, byte_range=rg, threads=threads, range_info=infos[i])
                   for i, rg in enumerate(ranges)]
        self.stats["shards"] = n
        yield from self._run(sources, jobs, pool, hm)
        # the seams: reader i must have stopped exactly where reader i + 1 found its first record (a FASTQ record start inside a
        # range is recognised by its framing -- a heuristic only there; a mismatch means records were lost or read twice)
        # (a range in which no record / no block starts reports -2 and drops out of the chain: its neighbours are adjacent)
        live = [(i, d) for i, d in enumerate(infos) if d.get("first", -2) != -2]
        if any("first" not in d for d in infos):
            raise RuntimeError(f"{filepath}: a shard's reader did not finish; rerun with one reader (--shards 1)")
        for (i, da), (j, db) in zip(live, live[1:]):
            if da.get("stopped") != db.get("first"):
                raise RuntimeError(f"{filepath}: shard {i} stopped at {da.get('stopped')}, shard {j} began at {db.get('first')}: the file cannot be read by "
                                   f"byte ranges (records with unusual line layout around the cut); rerun with one reader (--shards 1)")

    @staticmethod
    def _probe_records(hm):
        """auto mode: the file's first batch is a small probe (its verdict decides the mode of the rest; a whole 64-Mbase batch scanned
        in two passes for nothing cost a third of the 300 MB benchmark file's run)."""
        return PROBE_READS if hm and hm["auto"] and hm["bp"] else 0

    def _heads_mode(self, jobs, filepath=None):
        """(ask, hm): the reader's heads_bp for the next batch of one file (a callable: seqio.read_batches_packed asks before every
        batch) and the state behind it, which belongs to ONE pass over one file (it used to live on the pool, where two
        generators alive at once steered each other's mode).
        Heads mode needs step 1 in every job and the same no_bp; "auto" keeps it while the batches seen so far say that it pays:
        reads several times longer than the two heads, and few of them passing step 1 (real WGS input: < 1 % telomeric).  A
        telomere-enriched file (the demo, the synthetic benchmarks: every read passes) drops to the one-pass route after its
        first batch -- there the second pass would upload what the first one spared."""
        mode = self.two_pass
        if mode == "auto" and filepath is not None:
            try:
                small = os.path.getsize(filepath) < (AUTO_MIN_FILE_BYTES // 4 if str(filepath).endswith(".gz") else AUTO_MIN_FILE_BYTES)
            except OSError:
                small = False
            if small:
                mode = "off"
        no_bp = {int(j.prm.no_bp) for j in jobs}
        ok = mode in ("auto", "on") and len(no_bp) == 1 and all(j.prm.flags & hiplib.F_STEP1 for j in jobs) and min(no_bp) > 0
        hm = {"bp": min(no_bp) if ok else 0, "auto": mode == "auto", "asked": 0, "first": threading.Event()}

        def ask():
            # auto: the file opens with a small probe batch in heads mode (_probe_records); until the verdict on it is in, the reader
            # goes on with ordinary whole-read batches (valid in either mode: nothing waits), then with what the verdict says
            hm["asked"] += 1
            if hm["auto"] and hm["bp"] and hm["asked"] > 1 and not hm["first"].is_set():
                return 0
            return hm["bp"]
        return ask, hm

    def _heads_feedback(self, pb, outs, hm=None):
        """Called by the workers with every finished batch: the statistics, and auto mode's decision."""
        st = self.stats
        with self._stats_lock:
            st["batches"] += 1
            if not hasattr(pb, "n_bases"):              # (a list of records / an ASCII batch: scan_stream)
                if hm:
                    hm["first"].set()
                return
            st["input_bases"] += int(pb.n_bases)
            if getattr(pb, "full_len", None) is None:
                st["upload_bytes"] += int(getattr(pb, "uploaded_bytes", pb.n_bases * 3 // 8))
                if hm:
                    hm["first"].set()
                return
            st["heads_batches"] += 1
            full = np.asarray(pb.full_len, np.int64)
            frac = max((float(np.mean(o[0]["pass"] != 0)) if len(o[0]) else 0.0) for o in outs)
            st["upload_bytes"] += int(getattr(pb, "uploaded_bytes", 0))
            if hm and hm["auto"] and hm["bp"] and len(full) >= 8:
                if frac > TWO_PASS_MAX_PASSING or float(full.mean()) < 4 * hm["bp"]:
                    hm["bp"] = 0                            # the rest of the file goes up whole
            if hm:
                hm["first"].set()

    def _staging_pool(self, words_cap, reads_cap, sets=None):
        """The pinned staging buffers of this engine set: allocated ONCE per (engine set, geometry) and kept on the first engine,
        so that a new EnginePool per input file (main.process_file_multi) reuses them -- they are only freed with the context
        (a pool per file leaked ~28 MB of pinned memory per buffer set and file: ADVICE r2).  Every context uploads from them
        asynchronously (the library keeps a process-wide registry of pinned buffers)."""
        from . import seqio
        owner = self.engines[0]
        pools = getattr(owner, "_staging_pools", None)
        if pools is None:
            pools = owner._staging_pools = {}
        sets = sets or len(self.engines) + 2
        key = (words_cap, reads_cap, sets)
        if key not in pools:
            pools[key] = seqio.BufferPool(sets, words_cap, reads_cap, getattr(owner, "host_alloc", None))
        return pools[key]

    def scan_stream(self, records, prm, want_sums=False, want_raw=False, max_bases=None):
        return self._single(self._run(record_batches(records, max_bases=max_bases or BATCH_BASES), [Job(self.patterns, prm, want_sums, want_raw)]))

    @staticmethod
    def _single(it):
        for b, outs in it:
            yield (b,) + outs[0]

    # -- the pipeline
    def _run(self, batches, jobs, pool=None, hm=None):
        """`batches`: an iterator of batches, or a LIST of them (the shards of one file, in file order): one reader thread per source,
        one shared queue, results yielded source after source, batch after batch."""
        sources = list(batches) if isinstance(batches, (list, tuple)) else [batches]
        multi = len(sources) > 1
        if multi and any(getattr(j, "raw_sink", None) is not None or j.want_sums or j.want_raw for j in jobs):
            raise ValueError("several sources: per-read records only")
        n = len(self.engines)
        if pool is not None:
            pool.abort.clear()
        q_in: queue.Queue = queue.Queue(maxsize=n + 1)
        q_out: queue.Queue = queue.Queue()
        stop = threading.Event()
        # At most n + 3 batches between the reader and the consumer: a batch's results (raw rows: hundreds of MB per batch and k) wait
        # in `pending` until the consumer has dealt with the batches before it, and a consumer slower than the GPUs -- the CLI writing
        # raw rows -- let them pile up (configs[4]'s shard: 9.9 GB resident one second into the run).  The reader takes the tokens, in
        # batch order, so the batch the consumer waits for always has one.  (Several sources: per-read records only, a few MB per
        # batch -- the later shards' results simply wait for their turn.)
        in_flight = None if multi else threading.Semaphore(n + 3)
        readers_left = [len(sources)]
        rl_lock = threading.Lock()

        def reader(si, it):
            count = 0
            try:
                for b in it:
                    while in_flight is not None and not stop.is_set() and not in_flight.acquire(timeout=0.2):
                        pass
                    while not stop.is_set():
                        try:
                            q_in.put(((si, count), b), timeout=0.2)
                            break
                        except queue.Full:
                            continue
                    if stop.is_set():
                        if hasattr(b, "release"):
                            b.release()
                        break
                    count += 1
                q_out.put(("eof", si, count))
            except BaseException as e:          # parse errors surface in the consumer
                q_out.put(("error", None, e))
            finally:
                with rl_lock:
                    readers_left[0] -= 1
                    last = readers_left[0] == 0
                if last:
                    for _ in range(n):
                        q_in.put(None)

        def worker(eng):
            try:
                while True:
                    item = q_in.get()
                    if item is None:
                        return
                    key, b = item
                    outs = scan_jobs(eng, b, jobs, 0, seq=None if multi else key[1])
                    self._heads_feedback(b, outs, hm)
                    q_out.put(("batch", key, (b, outs)))
            except BaseException as e:
                stop.set()
                q_out.put(("error", None, e))

        threads = [threading.Thread(target=reader, args=(si, it), daemon=True) for si, it in enumerate(sources)] + \
                  [threading.Thread(target=worker, args=(e,), daemon=True) for e in self.engines]
        for t in threads:
            t.start()
        pending, totals, cur, nxt = {}, {}, 0, 0
        try:
            while cur < len(sources):
                if cur in totals and nxt >= totals[cur]:
                    cur, nxt = cur + 1, 0
                    continue
                if (cur, nxt) in pending:
                    yield pending.pop((cur, nxt))
                    nxt += 1
                    if in_flight is not None:
                        in_flight.release()            # (the consumer came back for more: it is done with that batch)
                    continue
                kind, key, payload = q_out.get()
                if kind == "error":
                    raise payload
                if kind == "eof":
                    totals[key] = payload
                    continue
                pending[key] = payload
        finally:
            stop.set()
            for job in jobs:
                if getattr(job, "raw_sink", None) is not None:
                    job.raw_sink.abort()               # a worker waiting for its turn in the raw-count file gives up (no-op after a clean run)
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
