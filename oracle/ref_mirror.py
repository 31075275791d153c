"""TEST / BASELINE INFRASTRUCTURE ONLY -- the reference's CPU path the way the reference runs it.

bench.py's `cpu_baseline.reference_python` leg times THIS on the GPU box's host cores (the reference's own files cannot
travel there): a pure-Python mirror of Topsicle's per-file worker, structured like the reference and using the same
primitives --

  * `re.finditer` per pattern over the first / reversed-last 1000 bases      (Topsicle/allsteps.py:167-198, patternTRC_count)
  * the file is parsed again for every passing read, both tails are windowed, `re.finditer` per window per pattern,
    `matches or 1`, mean over the patterns                                    (allsteps.py:257-297, bound_detect)
  * single-split l2 change-point with numpy `var()` per candidate, jump 5, min_size 2, tuple-max tie rule
                                                                              (allsteps.py:310-311 + ruptures 1.1.9 Binseg / CostL2)
  * one task per input FILE on a `multiprocessing.Pool`                      (Topsicle/main.py:232-235)

so a single file uses ONE core whatever --threads says; the bench splits its sample into one file per usable core, which is
what the reference's README (267-268) tells users to do.  The algorithmic restatement it shares with the oracle
(`topsicle_oracle.binseg_l2_numpy`, `kmer_table`) is pinned by tests/test_oracle_golden.py; tests/test_ref_mirror.py pins
this file's results against the reference-generated goldens.  Only tests/ and bench.py's cpu_baseline leg import it.
"""
from __future__ import annotations

import os
import re
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import topsicle_oracle as orc  # noqa: E402


def parse_fastq(path):
    """(id, sequence) of every record: the stand-in for Bio.SeqIO.parse (allsteps.py:142-146) on 4-line FASTQ."""
    with open(path, "rt") as h:
        while True:
            head = h.readline()
            if not head:
                return
            seq = h.readline().rstrip("\n")
            h.readline()
            h.readline()
            yield head[1:].split()[0], seq


def pattern_trc_count(path, motif, k, read_length, cutoff, no_bp=1000):
    """allsteps.py:152-204: [[id, best pattern, 'forward'|'reverse', trc]] of the reads over the cutoff."""
    pats = orc.kmer_table(motif, k)
    rx = [re.compile(p) for p in pats]
    ratio = no_bp / len(motif)
    out = []
    for rid, seq in parse_fastq(path):
        if len(seq) > read_length:
            s = seq[:no_bp].upper()
            e = seq[-no_bp:][::-1].upper()
            cs = [len(list(r.finditer(s))) / ratio for r in rx]
            ce = [len(list(r.finditer(e))) / ratio for r in rx]
            ms, me = max(cs), max(ce)
            if ms > me:
                tail, best, pat = "forward", ms, pats[cs.index(ms)]
            else:
                tail, best, pat = "reverse", me, pats[ce.index(me)]
            if best > cutoff:
                out.append([rid, pat, tail, best])
    return out


def bound_detect(path, read, pats, window, slide, trimfirst, maxlengthtelo, tail):
    """allsteps.py:227-338: re-parses the file, windows BOTH tails, keeps the chosen one, Binseg-l2 single split."""
    rx = [re.compile(p) for p in pats]
    res = []
    for rid, seq in parse_fastq(path):
        if rid != read:
            continue                                            # (no break upstream either: allsteps.py:257-259)
        m = min(maxlengthtelo, len(seq))
        tails = {"forward": seq[trimfirst:m].upper(), "reverse": seq[::-1].upper()[trimfirst:m]}
        means = {}
        for name, s in tails.items():
            ys = []
            for start in range(0, len(s) - window + 1, slide):
                w = s[start:start + window - 1]                # W - 1 characters (allsteps.py:219-221)
                ys.append(sum((len(list(r.finditer(w))) or 1) for r in rx) / len(rx))
            means[name] = ys
        y = np.asarray(means[tail], dtype=np.float64)
        bkp = orc.binseg_l2_numpy(y)[0] if len(y) else None
        boundary = 0
        if bkp is not None:
            b = bkp * slide + trimfirst
            boundary = b if 0 < b <= maxlengthtelo else 0
        res.append([read, boundary])
    return res


def process_file(path, motif, k, min_len, cutoff, window, slide, trimfirst, maxlengthtelo):
    """main.py:52-154 without the file outputs: step 1 over the file, then one bound_detect per passing read."""
    pats = orc.kmer_table(motif, k)
    rows = []
    for rid, _pat, tail, trc in pattern_trc_count(path, motif, k, min_len, cutoff):
        r = bound_detect(path, rid, pats, window, slide, trimfirst, maxlengthtelo, tail)
        rows.append((rid, tail, trc, r[0][1] if r else 0))
    return rows


def _worker(args):
    t0 = time.perf_counter()
    rows = process_file(*args)
    return len(rows), time.perf_counter() - t0


def timed_pool(paths, motif, k, min_len, cutoff, window, slide, trimfirst, maxlengthtelo, processes):
    """One task per file on a Pool (main.py:232-235).  Returns (wall seconds, reads that passed, per-file seconds)."""
    import multiprocessing as mp
    jobs = [(p, motif, k, min_len, cutoff, window, slide, trimfirst, maxlengthtelo) for p in paths]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(processes=processes) as pool:
        res = pool.map(_worker, jobs, chunksize=1)
    wall = time.perf_counter() - t0
    return wall, sum(r[0] for r in res), [r[1] for r in res]
