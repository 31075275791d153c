"""`topsicle` command line on MI355X: same flags, output files and log lines as the reference's
Topsicle/main.py (flags main.py:319-334; CSV header :200, rows :138; log lines :266, 293, 302,
309), with the per-read work done in batched HIP launches instead of the reference's
one-process-per-file pool and per-read file re-parsing (main.py:125-152, 232-235).

Extra flags (not in the reference): --gpus N shards batches of reads over N GPUs of the node
(one host thread + one context per GPU, no collective); --device picks the first GPU.
"""
from __future__ import annotations

import argparse
import csv
import datetime
import os
import queue
import sys
import threading
import time
from collections import defaultdict

import numpy as np

from . import allsteps, batch, hiplib, rawnpz, seqio

version_number = "1.0.0"
Topsicle_output_prefix = "Topsicle"
CONTEXTS_PER_GPU = 2
WRITER_THREADS = 1                # threads that write the passing records (native FASTQ writer, pwritev at offsets known up front).  Three were measured
                                  # (round 5, configs[3]'s shard: 7.5 GB of records): buffered writes into ONE file serialise on the inode, three threads
                                  # reach 1.4 - 2.0 GB/s each where one reaches 4.9 - 6.5 -- 1.68 -> 2.30 s (.gz), 1.50 -> 2.07 s (BGZF); one it is
LAST_TIMINGS = {}                 # seconds of the last analysis_run: reads (step 1 + 2 + outputs) / summary (fit + PNG); bench.py's e2e leg reads it
PLOT_THREADS = []                 # quadratic-fit PNGs being rendered off the critical path (summarize); wait_for_plots() joins them


def wait_for_plots():
    """Block until every summary PNG of earlier runs is on disk; an exception inside a drawing is reported here (it used to
    be lost in its thread and a missing quadfit_*.png went unnoticed)."""
    _PLOT_JOBS.join()
    errs, PLOT_ERRORS[:] = list(PLOT_ERRORS), []
    for path, err in errs:
        tprint(f"summary plot {path} failed: {err!r}")
    return errs


# The summary pictures are drawn by ONE worker thread, one after the other (one thread per k-mer length rendered them
# concurrently -- Agg rendering with shared font caches is only safe on recent matplotlib)
import queue as _queue  # noqa: E402

_PLOT_JOBS: "_queue.Queue" = _queue.Queue()
PLOT_ERRORS: list = []


def _plot_worker():
    while True:
        fn, fargs, path = _PLOT_JOBS.get()
        try:
            fn(*fargs)
        except BaseException as e:                                   # reported by wait_for_plots()
            PLOT_ERRORS.append((path, e))
        finally:
            _PLOT_JOBS.task_done()


def _submit_plot(fn, fargs, path):
    if not PLOT_THREADS:                       # the one worker, started with the first picture; a process that ends waits for its queue
        import atexit
        t = threading.Thread(target=_plot_worker, daemon=True)
        t.start()
        PLOT_THREADS.append(t)
        atexit.register(wait_for_plots)
    _PLOT_JOBS.put((fn, fargs, path))


def warm_up_plotting():
    """Import matplotlib's Agg machinery on a helper thread while the reads are being scanned (the import costs more than the
    drawing): by the time the summary wants its PNG the modules are there."""
    def imp():
        try:
            from matplotlib.backends.backend_agg import FigureCanvasAgg  # noqa: F401
            from matplotlib.figure import Figure  # noqa: F401
        except Exception:
            pass
    t = threading.Thread(target=imp, daemon=True)
    t.start()
    return t


def get_log_path(args):
    log_dir = getattr(args, "outputDir", ".")
    os.makedirs(log_dir, exist_ok=True)
    return os.path.join(log_dir, "topsicle_run.log")


def tprint(*args, **kwargs):
    """Timestamped print, mirrored into <outputDir>/topsicle_run.log (main.py:37-45)."""
    msg = " ".join(str(a) for a in args)
    now = datetime.datetime.now().strftime("%Y-%m-%d %H:%M:%S")
    line = f"[{now}] {msg}"
    print(line)
    if hasattr(tprint, "logfile"):
        with open(tprint.logfile, "a") as f:
            f.write(line + "\n")


def _formats(seq_loc: str):
    """(input format, filtered-file extension) by file name, as main.py:68-81 decides it."""
    if seq_loc.endswith(".gz"):
        fq = seq_loc.endswith(".fastq.gz") or seq_loc.endswith(".fq.gz")
    else:
        fq = seq_loc.endswith(".fastq") or seq_loc.endswith(".fq")
    return "fastq" if fq else "fasta"


_CSV_LOCK = __import__("threading").Lock()      # telolengths_all.csv is appended batch by batch from every file in flight


def process_file(args, seq_loc, telo_phrase, pattern, sliding_val, engines):
    """One input file, one k (main.py:52-154).  Returns [(file_name, telo_phrase, [[readID, telolen]], trc), ...] in read order."""
    return process_file_multi(args, seq_loc, [(telo_phrase, pattern, sliding_val)], engines)[0]


def process_file_multi(args, seq_loc, phrases, engines):
    """One input file, every requested k in ONE pass: each batch is parsed, packed and uploaded once and scanned once per
    k-mer table (`phrases` = [(telo_phrase, pattern list, slide), ...]).  Per k it does what the reference's process_file
    does (main.py:52-154): step 1 + filtered file + step 2 rows; the per-read work of a batch is numpy arithmetic on the
    kernel's result records, one bulk write of the passing records and one bulk CSV append.  Rows of the first k are
    appended to telolengths_all.csv batch by batch; the other k's rows are returned for the caller to append behind them
    (the reference writes all rows of one k before the next).  Returns one row list per k:
    [(file_name, telo_phrase, [[readID, telolen]], trc), ...] in read order."""
    tprint("subsetting raw dataset based on TRC cutoff")
    base_name = os.path.basename(seq_loc)
    file_name = os.path.splitext(base_name)[0]
    min_cutoff = min(args.cutoff) if isinstance(args.cutoff, (list, tuple)) else args.cutoff
    no_bp = 1000
    ratio = no_bp / len(args.pattern)
    want_sums = bool(args.plot)
    want_raw = bool(args.rawcountpattern)
    # --rawcountformat npz: one archive per input file and k, its rows written device -> file by the batch workers (rawnpz.py)
    npz = want_raw and getattr(args, "rawcountformat", "csv") == "npz"
    raw_npz = [rawnpz.RawNpzWriter(f"{args.outputDir}/rawcount_{telo_phrase}_{file_name}.npz", pattern, sliding_val, args.read_check or None)
               if npz else None for telo_phrase, pattern, sliding_val in phrases]
    jobs = [batch.Job(pattern, hiplib.make_params(
        no_bp=no_bp, min_len=args.minSeqLength, min_count=allsteps.min_count_for_cutoff(min_cutoff, ratio, no_bp),
        window=args.windowSize, slide=slide, trimfirst=args.trimfirst, maxlen=args.maxlengthtelo,
        flags=hiplib.F_STEP1 | hiplib.F_WINDOWS | hiplib.F_BINSEG), want_sums, want_raw, raw_sink=sink)
        for (_k, pattern, slide), sink in zip(phrases, raw_npz)]

    fmt = _formats(seq_loc)
    fasta_temp = os.path.join(args.outputDir, f"{file_name}_trc_over_{min_cutoff}.fasta")
    out_handle = None
    if os.path.exists(fasta_temp):
        tprint(f"Temporary fasta file already exists: {fasta_temp}. Using existing file.")
    else:
        fasta_temp = os.path.join(args.outputDir, f"{file_name}_trc_over_{min_cutoff}.{fmt}")
        out_handle = open(fasta_temp, "wb")
    # Which k's passing records the file holds after upstream's OUTER loop over k (main.py:206-235, 64-78): FASTA input -- the
    # first k writes <name>.fasta, every later k finds it and reuses it, so the FIRST k's records stay; FASTQ input -- only
    # `.fasta` is ever checked, the `.fastq` file is rewritten for every k, so the LAST k's records stay.
    writer_k = 0 if fmt == "fasta" else len(phrases) - 1

    rows = [[] for _ in phrases]
    image_num = [1] * len(phrases)
    csv_path = f"{args.outputDir}/telolengths_all.csv"
    pool = batch.EnginePool(engines, two_pass=getattr(args, "twopass", None))
    # the passing records are written by a helper thread while the next batches are scanned (the native writer releases the GIL:
    # pwritev straight from the mapped input).  Round 5: every batch whose records leave through the native writer gets its place in
    # the file up front (a record takes header + 2 x bases + 6 bytes), so the writes need no order -- several threads could write
    # different batches at once; measured, they do not pay (WRITER_THREADS above).  BASELINE configs[3]'s shard rewrites 7.5 GB: that
    # one thread at 5 - 6 GB/s is what the run waits for.  Batches of other kinds (FASTA, ASCII batches) are written in order, after
    # everything before them has landed.
    wq: "queue.Queue" = queue.Queue(maxsize=2 * WRITER_THREADS + 2)
    werr = []
    wstat = {"write_s": 0.0, "blocked_s": 0.0, "records": 0}             # where the filtered file's time goes (logged below)
    wlock = threading.Lock()
    out_off = [0]                                                         # bytes of the filtered file given out so far

    def writer():
        while True:
            item = wq.get()
            try:
                if item is None:
                    return
                if not werr:
                    t0 = time.perf_counter()
                    item[0].write_records(out_handle, item[1], fmt, offset=item[2])
                    with wlock:
                        wstat["write_s"] += time.perf_counter() - t0
                        wstat["records"] += len(item[1])
            except BaseException as e:                                     # surfaces in the main thread below
                werr.append(e)
            finally:
                wq.task_done()
    wthreads = [threading.Thread(target=writer, daemon=True) for _ in range(WRITER_THREADS)] if out_handle is not None else []
    for t in wthreads:
        t.start()
    wthread = wthreads[0] if wthreads else None

    def write_passing(pb, idx):
        nbytes = pb.native_fastq_bytes(idx, fmt)
        t0 = time.perf_counter()
        if nbytes is not None:
            wq.put((pb, idx, out_off[0]))
            out_off[0] += nbytes
        else:
            wq.join()                                  # an ordered write: everything before it has landed
            out_handle.seek(out_off[0])
            pb.write_records(out_handle, idx, fmt)
            out_handle.flush()
            out_off[0] = out_handle.tell()
            wstat["records"] += len(idx)
        wstat["blocked_s"] += time.perf_counter() - t0

    # one big plain file on several GPUs: cut into byte ranges, a reader team per range (batch.EnginePool.scan_file_jobs; per-read
    # records only -- plots and raw rows keep one reader).  --shards N forces it (tests); the default cuts files of at least 256 MiB per GPU
    n_shards = int(getattr(args, "shards", 0) or 0)
    shard_min = 4096
    if n_shards <= 0:
        n_shards, shard_min = max(1, len({getattr(e, "device", 0) for e in engines})), 256 << 20
    ok = False
    try:
        for pb, outs in pool.scan_file_jobs(seq_loc, jobs, shards=n_shards, shard_min_bytes=shard_min):
            for n, ((telo_phrase, pattern, sliding_val), (res, sums, raw, win_off)) in enumerate(zip(phrases, outs)):
                idx = np.nonzero(res["pass"])[0]
                if wthread is not None and len(idx) and n == writer_k:
                    write_passing(pb, idx)                                   # every passing record (main.py:83-86)
                ids = [pb.read_id(int(i)) for i in idx]
                if args.read_check:
                    keep = [j for j, rid in enumerate(ids) if rid == args.read_check]
                    idx, ids = idx[keep], [ids[j] for j in keep]
                if not len(idx):
                    continue
                r = res[idx]
                fwd = r["tail"] == 0
                trc = np.where(fwd, r["best_start"], r["best_end"]) / ratio  # float64, like int / float upstream
                lens = pb.desc["len"][idx].astype(np.int64)
                m = np.minimum(args.maxlengthtelo, lens)
                bkp = r["bkp"].astype(np.int64)
                point = np.where(bkp >= 0, bkp * sliding_val + args.trimfirst, 0)
                telolen = np.where((point <= m) & (point != 0), point, 0)    # allsteps.py:330-333
                for j in np.nonzero(bkp < 0)[0]:
                    tprint(f"read {ids[j]}: {int(r['n_win'][j])} windows, no admissible change point; reporting 0")
                telo_l, trc_l = telolen.tolist(), trc.tolist()
                if n == 0:
                    with _CSV_LOCK, open(csv_path, mode="a", newline="") as fh:
                        csv.writer(fh).writerows(zip([file_name] * len(ids), [telo_phrase] * len(ids), ["%.3f" % t for t in trc_l], ids, telo_l))
                rows[n] += [(file_name, telo_phrase, [[rid, tl]], t) for rid, tl, t in zip(ids, telo_l, trc_l)]
                if raw_npz[n] is not None:
                    # the rows are in the archive already (written by the worker that scanned the batch); what is kept here is
                    # one id, one tail and one window count per read -- no per-read loop over the rows
                    assert np.array_equal(raw.keep, idx)
                    raw_npz[n].add_reads(ids, np.where(fwd, "forward", "reverse").tolist(), r["n_win"])
                if args.plot or (args.rawcountpattern and raw_npz[n] is None):   # per-read artefacts (main.py:140-150)
                    for j, i in enumerate(idx):
                        tail = "forward" if fwd[j] else "reverse"
                        if args.plot and r["n_win"][j] > 0:
                            import matplotlib.pyplot as plt
                            y = sums[win_off[i]:win_off[i + 1]] / len(pattern)
                            allsteps._plot_changepoint(ids[j], y, sliding_val, args.trimfirst, int(point[j]), args.rangecp or int(m[j]))
                            plt.savefig(f"{args.outputDir}/plot_{telo_phrase}_{image_num[n] + j}.png", format="png", dpi=300)
                            plt.close()
                        if args.rawcountpattern and raw_npz[n] is None:
                            _write_rawcount(args, telo_phrase, image_num[n] + j, pattern, sliding_val, raw[win_off[i]:win_off[i + 1]], tail)
                image_num[n] += len(idx)
        ok = True
    finally:
        for _ in wthreads:
            wq.put(None)
        for t in wthreads:
            t.join()
        if out_handle is not None:
            out_handle.close()
        if not ok or werr:                                  # no half-written archive is left behind, whatever went wrong
            for w in raw_npz:
                if w is not None:
                    w.discard()
    if werr:
        raise werr[0]
    for w in raw_npz:
        if w is not None:
            w.finish()
    if out_handle is not None:
        tprint(f"Temporary fasta file with TRC more than {min_cutoff}:", fasta_temp)
        if wstat["write_s"] > 0.05:                 # (big outputs only: what bounds a run that rewrites most of its input)
            nbytes = os.path.getsize(fasta_temp)
            tprint(f"{base_name}: {wstat['records']} passing records, {nbytes} bytes written in {wstat['write_s']:.2f} thread-seconds on {len(wthreads)} writer threads "
                   f"({nbytes / wstat['write_s'] / 1e9:.2f} GB/s per thread); the scan loop waited {wstat['blocked_s']:.2f} s for them")
    st = pool.stats
    if st.get("shards", 0) > 1:
        tprint(f"{base_name}: read as {st['shards']} byte ranges, one reader team each")
    if st["heads_batches"]:
        tprint(f"{base_name}: {st['heads_batches']} of {st['batches']} batches scanned in two passes (read ends first, then the reads that pass): "
               f"{st['upload_bytes']} bytes uploaded for {st['input_bases']} bases")
    return rows


def _write_rawcount(args, telo_phrase, image_num, pattern, slide, block, tail):
    """rawcount_{k}_{i}.csv in the layout DataFrame.to_csv gives upstream (main.py:146-150)."""
    import pandas as pd
    n_win = block.shape[0]
    df = pd.DataFrame({
        "tail": tail,
        "position": np.repeat(np.arange(n_win, dtype=np.int64) * slide, len(pattern)),
        "pattern": np.tile(np.array(pattern, dtype=object), n_win),
        "count": block.reshape(-1).astype(np.int64),
    })
    df.to_csv(f"{args.outputDir}/rawcount_{telo_phrase}_{image_num}.csv")


def analysis_run(args, engines=None, engine_factory=None, wait_plots=True):
    """`engines`: contexts to use (tests inject theirs).  `engine_factory()` -> a fresh list of contexts: given (or by
    default on real GPUs), several input files are processed concurrently, one host thread + one set of contexts per
    file in flight -- the role of upstream's `Pool(num_cores)` over files (main.py:232-235)."""
    LAST_TIMINGS.clear()
    print("---- Topsicle run parameters ---")
    for k, v in vars(args).items():
        tprint(f"{k}: {v}")
    print("---------------------")
    tprint("Starting Topsicle analysis")
    os.makedirs(args.outputDir, exist_ok=True)

    if args.threads is not None:
        num_cores = args.threads
        tprint(f"Specified number of cores are/is: {num_cores}")
    else:
        num_cores = len(os.sched_getaffinity(0))
        tprint(f"By default, Topsicle allocates nmber of cores: {num_cores}")

    output_csv = f"{args.outputDir}/telolengths_all.csv"
    tprint(f"Output will be here: {output_csv}")
    if os.path.exists(output_csv) and os.path.getsize(output_csv) > 0:
        if args.override:
            tprint(f"Output file {output_csv} already exists and will be overridden becuz having --override flag.")
            os.remove(output_csv)
        else:
            tprint(f"Output file {output_csv} already exists and is not empty. Exiting to avoid overwrite. Use --override to force overwrite.")
            sys.exit(1)

    if args.telophrase is None:
        telo_phrases = [len(args.pattern) - 2]
        tprint(f"No telophrase provided, use kmer: {telo_phrases}")
    else:
        telo_phrases = args.telophrase if isinstance(args.telophrase, list) else [args.telophrase]
    print("---------------------")

    with open(output_csv, mode="w", newline="") as fh:
        csv.writer(fh).writerow(["file_number", "phrase", "trc", "readID", "telo_length"])

    if engines is None:
        n_gpus = max(1, getattr(args, "gpus", 1) or 1)
        first = getattr(args, "device", 0) or 0
        if engine_factory is None:
            def engine_factory():
                # two contexts (launch queues) per GPU: the upload / launch ramp / download of one batch overlaps the scan of another
                return [hiplib.HipScanner(first + i // CONTEXTS_PER_GPU) for i in range(n_gpus * CONTEXTS_PER_GPU)]
        engines = engine_factory()
        tprint(f"GPU engines: {[e.device_info() for e in engines]}")

    phrase_to_telo = defaultdict(list)
    phrase_to_trc = defaultdict(list)
    phrases = []
    for telo_phrase in telo_phrases:
        if telo_phrase > len(args.pattern):
            tprint("Cannot have length of subset larger than length of pattern")
            tprint(f"Cannot get {telo_phrase}-bp cut from {len(args.pattern)}-bp pattern")
            sys.exit()
        sliding_val = args.slide if args.slide else len(args.pattern)
        pattern = allsteps.patterns_to_search(telopattern=args.pattern, cut_length=telo_phrase)
        tprint("patterns to search:", pattern)
        phrases.append((telo_phrase, pattern, sliding_val))

    filenames = []
    if os.path.isdir(args.inputDir):
        for root, _dirs, files in os.walk(args.inputDir):
            for filename in files:
                filenames.append(os.path.join(root, filename))
    else:
        filenames.append(args.inputDir)

    # every k in ONE pass over the input (the reference loops over k outermost and re-parses every file per k,
    # main.py:206-235): per file a list of row lists, one per k
    tprint("begin processing reads")
    t_reads = time.perf_counter()
    per_file = _process_files(args, filenames, phrases, engines, engine_factory, num_cores)
    LAST_TIMINGS["reads_s"] = time.perf_counter() - t_reads
    tprint("finished processing all reads")
    print("---------------------")
    for n, (telo_phrase, _pattern, _slide) in enumerate(phrases):
        late = []
        for file_rows in per_file:
            for entry in file_rows[n]:
                phrase_to_telo[entry[1]].append(float(entry[2][0][1]))
                phrase_to_trc[entry[1]].append(float(entry[3]))
                if n > 0:
                    late.append([entry[0], entry[1], "%.3f" % entry[3], entry[2][0][0], entry[2][0][1]])
        if late:                                   # rows of the later k's follow the first k's, in upstream's order
            with _CSV_LOCK, open(output_csv, mode="a", newline="") as fh:
                csv.writer(fh).writerows(late)

    t_sum = time.perf_counter()
    summarize(args, phrase_to_telo, phrase_to_trc)
    LAST_TIMINGS["summary_s"] = time.perf_counter() - t_sum
    tprint("All telomere found, have a nice day.")
    if wait_plots:
        wait_for_plots()


def _process_files(args, filenames, phrases, engines, engine_factory, num_cores):
    """All input files: sequentially on the given contexts, or -- several files, a factory for more contexts, no per-read
    plots (pyplot is not thread-safe) -- up to min(num_cores, 8) files at a time, each on its own thread and contexts
    (parsing / gunzip is the bottleneck, the GPUs are shared).  Results keep file order: one list of per-k row lists per file."""
    workers = min(len(filenames), max(1, num_cores), 8)
    # (per-read plots and per-read CSVs go through pyplot / pandas on the calling thread; the npz archive is written natively by
    # the batch workers, one archive per file: nothing is shared between files)
    per_read_files = args.plot or (args.rawcountpattern and getattr(args, "rawcountformat", "csv") != "npz")
    if workers <= 1 or engine_factory is None or per_read_files:
        return [process_file_multi(args, seq_loc, phrases, engines) for seq_loc in filenames]
    import threading
    from concurrent.futures import ThreadPoolExecutor
    local = threading.local()
    made = []

    def work(seq_loc):
        if not hasattr(local, "engines"):
            local.engines = engine_factory()
            made.append(local.engines)
        return process_file_multi(args, seq_loc, phrases, local.engines)

    tprint(f"processing {len(filenames)} files, {workers} at a time")
    try:
        with ThreadPoolExecutor(max_workers=workers) as ex:
            return list(ex.map(work, filenames))
    finally:
        for engs in made:
            for e in engs:
                e.close()


def recommend_cutoff(vertex_x, max_trc, median_trc, inputtrc):
    """The cutoff advice of main.py:276-293 as a pure function: (recommended cutoff, log lines).  The fit's vertex is
    accepted unless it lies beyond the data (then the median TRC, or 0.9 if that is >= 1) or is implausibly low (< 0.4:
    the input cutoff takes over when it is higher)."""
    notes = []
    if vertex_x > max_trc:
        notes.append(f"Asymptotic TRC {vertex_x:.3f} is greater than max TRC, which is not expected. See plot.")
        use_median = median_trc < 1.0
        notes.append(f"Using median TRC value ({median_trc:.3f}) as asymptotic TRC instead." if use_median
                     else "Using 0.9 as asymptotic TRC instead, since asymptotic is greater than 1.0.")
        vertex_x = median_trc if use_median else 0.9
    if vertex_x < 0.4:
        notes.append("Quadratic fit suggests asymptotic TRC less than 0.4. See plot with fit line")
        if max_trc < 0.4:
            notes.append(f"Maximum TRC value in data is {max_trc:.3f}, which is less than 0.4, indicating low confidence in telomere detection.")
        if vertex_x < inputtrc:
            notes.append(f"Asymptotic TRC {vertex_x:.3f} is less than input cutoff {inputtrc:.3f}. Topsicle declares input TRC (={inputtrc}) as asymptotic TRC.")
            vertex_x = inputtrc
    return vertex_x, notes


def summarize(args, phrase_to_telo, phrase_to_trc):
    """Per-k medians and the quadratic-fit cutoff advice (main.py:248-306)."""
    inputtrc = args.cutoff[0] if isinstance(args.cutoff, (list, tuple)) else args.cutoff
    for phrase in sorted(phrase_to_telo):
        telo = np.asarray(phrase_to_telo[phrase], dtype=np.float64)
        trc = np.asarray(phrase_to_trc[phrase], dtype=np.float64)
        tprint(f"k-mer: {phrase}, with TRC >= {inputtrc}, median telomere length is {np.median(telo):.2f} bp")
        if len(telo) < 3:
            tprint("Not enough data points to recommend TRC cutoff.")
            continue
        median_trc = np.median(trc)
        fit_x, fit_y, coeffs = allsteps.fit_quadratic_and_find_vertex(list(trc), list(telo), inputtrc=inputtrc, median_trc=median_trc)
        # the picture is drawn on a helper thread: the numbers below do not wait for it (0.2 s of a 0.4 s run on 10 000 reads)
        png = os.path.join(args.outputDir, f"quadfit_{phrase}mer_{args.pattern}.png")
        _submit_plot(allsteps.plot_quadratic_fit, (trc, telo, coeffs, fit_x, fit_y, png), png)
        cutoff, notes = recommend_cutoff(fit_x, float(trc.max()), median_trc, inputtrc)
        for line in notes:
            tprint(line)
        tprint(f"asymptotic TRC, or recommended cutoff: {cutoff:.3f}")
        kept = telo[trc >= cutoff]
        if len(kept):
            tprint(f"Median telomere length for reads with TRC cutoff >= {cutoff:.3f}: {np.median(kept):.2f} bp")
        else:
            tprint(f"No read has TRC >= {cutoff:.3f}, please double check the data or submit log to GitHub.")


def build_parser():
    parser = argparse.ArgumentParser(description="Topsicle - Telomere length estimation from long reads (MI355X build)",
                                     formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("--inputDir", "-i", type=str, metavar="FILE/FOLDER", help="Required, Path to the input file or directory", required=True)
    parser.add_argument("--outputDir", "-o", type=str, metavar="FOLDER", help="Required, Path to the output directory", required=True)
    parser.add_argument("--pattern", metavar="CHAR", type=str, help="Required, Telomere repeat sequence (in 5' to 3' orientation). For e.g., in human use CCCTAA", required=True)
    parser.add_argument("--minSeqLength", metavar="INT", type=int, help="Minimum length of a long read sequence that will be analyzed", default=9000)
    parser.add_argument("--rawcountpattern", action="store_true", help="Output raw count of the k-mer for each window")
    parser.add_argument("--rawcountformat", choices=["csv", "npz"], default="csv",
                        help="(extension) csv: one rawcount_{k}_{i}.csv per read like upstream; npz: one columnar rawcount_{k}_{file}.npz per input file and k")
    parser.add_argument("--telophrase", nargs="+", metavar="INT", type=int, help="Length of telomere k-mer to search. By default will use telomere k-mer length minus 2")
    parser.add_argument("--cutoff", nargs="+", metavar="FLOAT", type=float, help="TRC statistics threshold", default=0.7)
    parser.add_argument("--windowSize", metavar="INT", type=int, help="Sliding window size", default=100)
    parser.add_argument("--slide", metavar="INT", type=int, help="Window sliding step. Default is telomere k-mer length")
    parser.add_argument("--trimfirst", metavar="INT", type=int, help="Length of intial number of base pairs to trim", default=100)
    parser.add_argument("--maxlengthtelo", metavar="INT", type=int, help="Longest possible length of telomere for any given read", default=20000)
    parser.add_argument("--plot", action="store_true", help="Optional, generate plot showing for each telomere read the abundance across the sequencing reead and the changepoint")
    parser.add_argument("--rangecp", metavar="INT", type=int, help="Optional, set range of changepoint plot for visualization, default is maxlengthtelo")
    parser.add_argument("--read_check", metavar="STR", type=str, help="Optional, get telomere of a specific read")
    parser.add_argument("--override", "-ov", action="store_true", help="Override telolengths_all.csv file but keep subset fastq")
    parser.add_argument("--threads", "-t", metavar="INT", type=int, help="Number of CPU cores to use (by default, all available cores)", default=None)
    # MI355X build only
    parser.add_argument("--gpus", metavar="INT", type=int, default=1, help="GPUs of this node to shard reads over")
    parser.add_argument("--device", metavar="INT", type=int, default=0, help="index of the first GPU to use")
    parser.add_argument("--shards", metavar="INT", type=int, default=0,
                        help="MI355X build: cut ONE plain input file into this many byte ranges, each decoded by a reader thread team of its own "
                             "(0 = one per GPU when the file has at least 256 MiB per GPU; compressed files keep one reader)")
    parser.add_argument("--twopass", choices=["auto", "on", "off"], default=None,
                        help="MI355X build: upload only the two 1000-base ends of every read for the TRC filter and the scanned part of the "
                             "reads that pass afterwards (auto, the default unless $TOPSICLE_TWO_PASS says otherwise: while few reads of a batch pass, "
                             "as in whole-genome read sets)")
    return parser


def main(argv=None):
    start_time = time.time()
    args = build_parser().parse_args(argv)
    tprint.logfile = get_log_path(args)
    warm_up_plotting()
    analysis_run(args, wait_plots=False)
    # the elapsed time is that of the ANALYSIS: the summary PNGs may still be rendering on the plot thread -- the process waits
    # for them when it ends (atexit), a caller of main() can call wait_for_plots() itself
    print(f"Elapsed time(s): {time.time() - start_time:.2f} seconds")


if __name__ == "__main__":
    main()
