"""End-to-end (parse + pack + PCIe + scan + output) rates of the path, measured on files on disk.

`bench.py` reports these beside `value` (which, by its contract, starts with the batch resident in HBM); they are what a
user of the `topsicle` CLI waits for.  Three legs over the SAME reads as the bench workload, written once as a plain
4-line FASTQ file (page-cache hot, like a file that was just produced by a basecaller):

  reader          native decode + 2-bit pack into pinned upload buffers, no GPU work                (bases/s)
  upload_scan     packed batches in pinned host memory -> upload -> fused scan -> results, one context          (bases/s)
  file_to_results batch.EnginePool.scan_file: reader -> upload (3 bits/base) -> fused scan -> results,
                  two contexts per GPU pulling from one queue                                        (bases/s)
  cli             `topsicle` itself (topsicle_amd.main): the above + filtered FASTQ + telolengths_all.csv + run summary
                  (replaces Topsicle/main.py:52-154, 156-309)                                        (bases/s)
(+ the same file as two-line and as 60-column FASTA, as ordinary gzip and as BGZF.)
Every leg reports the rate of its MEDIAN run as `value` (seconds_median / seconds_mean / seconds_best beside it).
"""
from __future__ import annotations

import os
import shutil
import tempfile
import time

import numpy as np


def write_fastq(path: str, bases: np.ndarray, offsets: np.ndarray, prefix: bytes = b"read", random_quality_seed: int | None = None):
    """Plain 4-line FASTQ.  Quality lines: all 'I' (compresses 6.7 : 1 with the bases) or, with a seed, ONT-like noise -- Phred
    3 .. 40, a clipped normal around 18: what a basecaller writes is about that incompressible (gzip -1: 1.9 : 1)."""
    raw = bases.tobytes()
    n = len(offsets) - 1
    qual_cache = {}
    qraw = None
    if random_quality_seed is not None:
        rng = np.random.default_rng(random_quality_seed)
        qraw = (np.clip(rng.normal(18.0, 7.0, int(offsets[-1])), 3, 40).astype(np.uint8) + 33).tobytes()
    with open(path, "wb", buffering=1 << 22) as h:
        for i in range(n):
            lo, hi = int(offsets[i]), int(offsets[i + 1])
            if qraw is not None:
                q = qraw[lo:hi]
            else:
                q = qual_cache.get(hi - lo)
            if q is None:
                q = qual_cache[hi - lo] = b"I" * (hi - lo)
            h.write(b"@%s%d\n" % (prefix, i))
            h.write(raw[lo:hi])
            h.write(b"\n+\n")
            h.write(q)
            h.write(b"\n")


def write_bgzf(path: str, src_path: str, block: int = 65280, level: int = 1):
    """bgzip-compatible copy of a file: independent gzip members of <= 64 KiB with the 'BC' extra field (block size - 1), then
    the empty end-of-file block (what `bgzip` writes; blocks inflate independently)."""
    import struct
    import zlib

    def member(chunk):
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        body = co.compress(chunk) + co.flush()
        bsize = 12 + 6 + len(body) + 8
        return (b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1) +
                body + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
    with open(src_path, "rb") as src, open(path, "wb", buffering=1 << 22) as dst:
        while True:
            chunk = src.read(block)
            if not chunk:
                break
            dst.write(member(chunk))
        dst.write(member(b""))


def _leg(times, n_bases, **extra) -> dict:
    """One end-to-end leg: `value` is the rate of the MEDIAN run (a best-of-3 on a 3 - 90 ms run of a page-cache-hot file
    is an upper bound, not a rate); the best and the mean are beside it."""
    t = sorted(float(x) for x in times)
    med = t[len(t) // 2] if len(t) % 2 else 0.5 * (t[len(t) // 2 - 1] + t[len(t) // 2])
    d = {"value": n_bases / med, "unit": "bases/s", "seconds_median": round(med, 4), "seconds_mean": round(float(np.mean(t)), 4),
         "seconds_best": round(t[0], 4), "runs": len(t)}
    d.update(extra)
    return d


def measure(bases, offsets, motif: str, k: int, slide: int, device: int = 0, contexts_per_gpu: int = 2, repeats: int = 5,
            workdir: str | None = None, with_cli: bool = True) -> dict:
    from . import allsteps, batch, hiplib, main as cli, seqio
    n_bases = int(offsets[-1])
    n_reads = len(offsets) - 1
    tmp = tempfile.mkdtemp(prefix="tps_e2e_", dir=workdir)
    out = {"reads": n_reads, "bases": n_bases, "contexts_per_gpu": contexts_per_gpu}
    try:
        fq = os.path.join(tmp, "reads.fastq")
        t0 = time.perf_counter()
        write_fastq(fq, bases, offsets)
        out["fastq_bytes"] = os.path.getsize(fq)
        out["fastq_write_s"] = round(time.perf_counter() - t0, 3)
        pats = allsteps.patterns_to_search(motif, k)
        engines = [hiplib.HipScanner(device) for _ in range(contexts_per_gpu)]
        try:
            # -- reader alone (pinned buffers, no GPU work)
            pool = seqio.BufferPool(4, batch.BATCH_BASES // 16, min(batch.BATCH_READS, batch.BATCH_BASES // 64), engines[0].host_alloc)
            times = []
            for _ in range(repeats + 1):
                t0 = time.perf_counter()
                nb = 0
                for pb in seqio.read_batches_packed(fq, pool):
                    nb += pb.n_bases
                    pb.release()
                times.append(time.perf_counter() - t0)
                assert nb == n_bases
            out["reader"] = _leg(times[1:], n_bases)           # (the first pass pins the pool)
            prm = hiplib.make_params(no_bp=1000, min_len=9000, min_count=allsteps.min_count_for_cutoff(0.7, 1000 / len(motif), 1000),
                                     window=100, slide=slide, trimfirst=100, maxlen=20000)
            # -- PCIe-inclusive: packed batches already in pinned host memory -> upload -> scan -> per-read results
            # (a pool of their own, one set more than the batches they can be: the reader asks for a set before it learns the file has ended)
            held_pool = seqio.BufferPool(6, batch.BATCH_BASES // 16, min(batch.BATCH_READS, batch.BATCH_BASES // 64), engines[0].host_alloc) if n_bases <= 3 * batch.BATCH_BASES else None
            held = list(seqio.read_batches_packed(fq, held_pool)) if held_pool is not None else []
            assert len(held) < 6
            if held:
                engines[0].set_patterns(pats)
                times = []
                for _ in range(repeats + 1):
                    t0 = time.perf_counter()
                    for pb in held:
                        engines[0].upload_packed(0, pb.seq2, pb.inv if pb.any_invalid else None, pb.desc)
                        engines[0].scan(0, prm)
                        engines[0].sync()
                        engines[0].results(0)
                    times.append(time.perf_counter() - t0)
                out["upload_scan"] = _leg(times[1:], n_bases, batches=len(held),
                                          note="one context, no overlap between batches: H2D of 3 bits per base + scan + D2H of the result rows")
                for pb in held:
                    pb.release()
            del held, pool, held_pool
            # -- file -> results
            ep = batch.EnginePool(engines, pats)
            prm = hiplib.make_params(no_bp=1000, min_len=9000, min_count=allsteps.min_count_for_cutoff(0.7, 1000 / len(motif), 1000),
                                     window=100, slide=slide, trimfirst=100, maxlen=20000)
            times = []
            for _ in range(repeats + 1):
                t0 = time.perf_counter()
                nr = npass = 0
                for pb, res, _s, _r, _w in ep.scan_file(fq, prm):
                    nr += pb.n
                    npass += int(res["pass"].sum())
                times.append(time.perf_counter() - t0)
                assert nr == n_reads
            times = times[1:]                           # the first pass allocates the pinned pool and the device buffers
            out["file_to_results"] = _leg(times, n_bases, reads_passing=npass, batch_bases=batch.BATCH_BASES)
            # -- the same reads as two-line FASTA (what a read set converted from FASTQ looks like): packed from the mapped file by
            # the same thread team (multi-line FASTA goes through the streaming decoder)
            fa = os.path.join(tmp, "reads.fasta")
            raw = bases.tobytes()
            with open(fa, "wb", buffering=1 << 22) as h:
                for i in range(n_reads):
                    h.write(b">read%d\n" % i)
                    h.write(raw[int(offsets[i]):int(offsets[i + 1])])
                    h.write(b"\n")
            times = []
            for _ in range(repeats):
                t0 = time.perf_counter()
                nr = 0
                for pb, res, _s, _r, _w in ep.scan_file(fa, prm):
                    nr += pb.n
                times.append(time.perf_counter() - t0)
                assert nr == n_reads
            out["fasta_file_to_results"] = _leg(times, n_bases, fasta_bytes=os.path.getsize(fa))
            os.unlink(fa)
            # -- and wrapped at 60 columns (what most FASTA files look like; Bio.SeqIO writes them so): joined line by line and
            # packed by the same thread team since round 4 (the one-thread streaming decoder before)
            with open(fa, "wb", buffering=1 << 22) as h:
                for i in range(n_reads):
                    h.write(b">read%d\n" % i)
                    sq = raw[int(offsets[i]):int(offsets[i + 1])]
                    h.write(b"\n".join(sq[j:j + 60] for j in range(0, len(sq), 60)))
                    h.write(b"\n")
            times = []
            for _ in range(repeats):
                t0 = time.perf_counter()
                nr = 0
                for pb, res, _s, _r, _w in ep.scan_file(fa, prm):
                    nr += pb.n
                times.append(time.perf_counter() - t0)
                assert nr == n_reads
            out["fasta_wrapped_file_to_results"] = _leg(times, n_bases, fasta_bytes=os.path.getsize(fa), columns=60)
            os.unlink(fa)
            # -- the same through ordinary gzip (one deflate stream: what `gzip` / `pigz` write, the reference's demo input): the
            # native reader inflates it with the thread team (csrc/tps_gzpar.h) instead of one zlib stream.  Twice: the file as
            # it is (constant quality lines: long matches, the inflater's easy case) and with ONT-like noisy quality lines
            # (half of the text nearly incompressible: literals and 3-byte matches, the inflater's hard case -- and the real one)
            import zlib

            def gz_leg(src_path, note):
                gz = src_path + ".gz"
                t0 = time.perf_counter()
                co = zlib.compressobj(1, zlib.DEFLATED, 31)
                with open(src_path, "rb") as src, open(gz, "wb") as dst:
                    while True:
                        blk = src.read(16 << 20)
                        if not blk:
                            break
                        dst.write(co.compress(blk))
                    dst.write(co.flush())
                t_gz = time.perf_counter() - t0
                legs = {}
                for name, env in (("parallel", None), ("zlib_stream", "1")):
                    if env:
                        seqio.io_option("no_pargz", int(env))
                    try:
                        times = []
                        for _ in range(repeats if env is None else 1):
                            t0 = time.perf_counter()
                            nr = 0
                            for pb, res, _s, _r, _w in ep.scan_file(gz, prm):
                                nr += pb.n
                            times.append(time.perf_counter() - t0)
                            assert nr == n_reads
                        legs[name] = times
                    finally:
                        seqio.io_option("no_pargz", 0)
                leg = _leg(legs["parallel"], n_bases, zlib_stream_value=n_bases / legs["zlib_stream"][0],
                           zlib_stream_seconds=round(legs["zlib_stream"][0], 4), text_bytes=os.path.getsize(src_path),
                           gz_bytes=os.path.getsize(gz), gz_write_s=round(t_gz, 2), note=note)
                os.unlink(gz)
                return leg

            out["gz_file_to_results"] = gz_leg(fq, "ordinary single-stream gzip (level 1) of the same FASTQ file -> results; zlib_stream_* = "
                                                   "the same with the reader's one-stream zlib path (reader option no_pargz)")
            fq_noisy = os.path.join(tmp, "reads_noisy_quality.fastq")
            write_fastq(fq_noisy, bases, offsets, random_quality_seed=1)
            out["gz_noisy_quality_file_to_results"] = gz_leg(fq_noisy, "the same reads with ONT-like noisy quality lines (Phred 3-40): gzip -1 leaves "
                                                                       "1.9 : 1, the inflater decodes literals and 3-byte matches")
            # -- and as bgzip writes it (BGZF: <= 64 KiB blocks that inflate independently), the noisy-quality file
            bg = fq_noisy + ".bgz.gz"
            t0 = time.perf_counter()
            write_bgzf(bg, fq_noisy)
            t_bg = time.perf_counter() - t0
            times = []
            for _ in range(repeats):
                t0 = time.perf_counter()
                nr = 0
                for pb, res, _s, _r, _w in ep.scan_file(bg, prm):
                    nr += pb.n
                times.append(time.perf_counter() - t0)
                assert nr == n_reads
            out["bgzf_noisy_quality_file_to_results"] = _leg(times, n_bases, bgzf_bytes=os.path.getsize(bg), bgzf_write_s=round(t_bg, 2),
                                                             note="the noisy-quality file as bgzip writes it (independent <= 64 KiB blocks, level 1): "
                                                                  "blocks inflate in parallel through the in-tree inflater")
            os.unlink(bg)
            os.unlink(fq_noisy)
        finally:
            for e in engines:
                e.close()
        # -- the CLI
        if with_cli:
            od = os.path.join(tmp, "out")
            argv = ["--inputDir", fq, "--outputDir", od, "--pattern", motif, "--telophrase", str(k), "--slide", str(slide), "--device", str(device)]
            times, parts = [], {}
            for r in range(repeats):
                shutil.rmtree(od, ignore_errors=True)
                t0 = time.perf_counter()
                _quiet(cli.main, argv)
                times.append(time.perf_counter() - t0)
                t1 = time.perf_counter()
                cli.wait_for_plots()                  # the summary PNG is drawn on a helper thread, behind the last log line
                plot_wait = time.perf_counter() - t1
                if times[-1] == min(times):
                    parts = {k: round(v, 4) for k, v in cli.LAST_TIMINGS.items()}
            rows = sum(1 for _ in open(os.path.join(od, "telolengths_all.csv"))) - 1
            out["cli"] = _leg(times, n_bases, seconds_split_of_best_run=parts,
                              reads_part_value=(n_bases / parts["reads_s"]) if parts.get("reads_s") else None,
                              csv_rows=rows, filtered_fastq_bytes=sum(os.path.getsize(os.path.join(od, f)) for f in os.listdir(od) if "_trc_over_" in f),
                              summary_png_finished_after_s=round(plot_wait, 4),
                              note="includes writing every passing record back out (all reads are telomeric in this workload) and the run summary; the summary's "
                                   "quadratic-fit PNG is drawn by a helper thread and lands summary_png_finished_after_s after the CLI's last line")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def measure_wgs(motif: str = "CCCTAA", k: int = 4, slide: int = 6, n_reads: int = 6000, read_len: int = 30000, telomeric_fraction: float = 0.01,
                device: int = 0, contexts_per_gpu: int = 2, repeats: int = 9, workdir: str | None = None) -> dict:
    """The step-1-dominated regime end to end: a FASTQ file of `read_len`-base reads of which
    `telomeric_fraction` are telomeric, file -> results with the one-pass upload (every read whole: 3 bits per base) and with the
    two-pass route (batch.scan_jobs_heads: the two 1000-base ends of every read, then the scanned part of the reads that pass).
    Reports both rates (median run) and the bytes that crossed PCIe per input base."""
    from . import allsteps, batch, hiplib, synth
    tmp = tempfile.mkdtemp(prefix="tps_e2e_wgs_", dir=workdir)
    out = {"reads": n_reads, "read_len": read_len, "telomeric_fraction": telomeric_fraction}
    try:
        bases, offsets, _ = synth.make_reads(n_reads, read_len, motif, seed=20250919 + 33, telomeric_fraction=telomeric_fraction)
        n_bases = int(offsets[-1])
        fq = os.path.join(tmp, "wgs.fastq")
        write_fastq(fq, bases, offsets)
        del bases
        pats = allsteps.patterns_to_search(motif, k)
        prm = hiplib.make_params(no_bp=1000, min_len=9000, min_count=allsteps.min_count_for_cutoff(0.7, 1000 / len(motif), 1000),
                                 window=100, slide=slide, trimfirst=100, maxlen=20000)
        engines = [hiplib.HipScanner(device) for _ in range(contexts_per_gpu)]
        try:
            rows = {}
            for mode in ("off", "on"):           # ("auto" probes files of 1 GiB and more: this one is smaller)
                ep = batch.EnginePool(engines, pats, two_pass=mode)
                times = []
                for rep in range(repeats + 1):
                    for key in ep.stats:
                        ep.stats[key] = 0
                    t0 = time.perf_counter()
                    got = []
                    for pb, res, _s, _r, _w in ep.scan_file(fq, prm):
                        got.append(res[["pass", "tail", "best_start", "best_end", "n_win", "bkp"]].copy())
                    times.append(time.perf_counter() - t0)
                rows[mode] = np.concatenate(got)
                out["two_pass_" + mode] = _leg(times[1:], n_bases, upload_bytes_per_input_base=ep.stats["upload_bytes"] / n_bases,
                                               heads_batches=ep.stats["heads_batches"], batches=ep.stats["batches"],
                                               reads_passing=int(rows[mode]["pass"].sum()))
            out["rows_equal"] = bool(np.array_equal(rows["off"], rows["on"]))
        finally:
            for e in engines:
                e.close()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def _quiet(fn, argv):
    """Run the CLI with its stdout chatter sent to /dev/null (the log file still gets everything)."""
    import contextlib
    with open(os.devnull, "w") as dn, contextlib.redirect_stdout(dn):
        fn(argv)


if __name__ == "__main__":
    import json
    import sys
    from . import synth
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    b, o, _ = synth.make_reads(n, 15000, "CCCTAA", seed=20250920)
    print(json.dumps(measure(b, o, "CCCTAA", 4, 6), indent=1))
