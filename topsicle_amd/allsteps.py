"""Host-side mirror of Topsicle/allsteps.py: same function names, argument meaning and return
shapes, with the counting and change-point work done by the HIP kernels (libtopsicle_hip.so).

Reference signatures kept: pattern_scramble_telo, patterns_to_search,
check_file_type, unzip_file, patternTRC_count, seq_cut_windows, bound_detect, rawCountPattern,
fit_quadratic_and_find_vertex.  The per-read functions exist for drop-in use and parity tests;
throughput runs go through topsicle_amd.batch (one launch per batch of reads).

The GPU engine is created lazily on first use (device 0 unless TOPSICLE_DEVICE is set).  If
the HIP library or the GPU is missing the call raises -- nothing is computed on the CPU.
"""
from __future__ import annotations

import logging
import os

import numpy as np

from . import hiplib, seqio
from .seqio import check_file_type  # noqa: F401  (part of the mirrored API)

logging.basicConfig(level=logging.ERROR)
version_number = "1.0.0"

_COMPLEMENT = str.maketrans("ACGT", "TGCA")
_engine = None


class BadSegmentationParameters(Exception):
    """Raised where ruptures raises it upstream: fewer than 7 windows (allsteps.py:310-311)."""


def get_engine():
    global _engine
    if _engine is None:
        _engine = hiplib.HipScanner(int(os.environ.get("TOPSICLE_DEVICE", "0")))
    return _engine


def set_engine(engine):
    """Install an engine object with HipScanner's interface (tests inject theirs here)."""
    global _engine
    _engine = engine


# ---------------------------------------------------------------------------- a1 / a2
def pattern_scramble_telo(pattern, cut_length):
    """Sorted distinct k-mers of pattern+pattern for every k in cut_length (allsteps.py:57-82)."""
    ks = cut_length if isinstance(cut_length, list) else [cut_length]
    doubled = (pattern + pattern).upper()
    found = set()
    for k in ks:
        found.update(doubled[i:i + k] for i in range(len(doubled) - k + 1))
    return sorted(found)


def patterns_to_search(telopattern, cut_length):
    """Pattern list in reference order: sorted k-mers, then their complements (no reversal);
    a list is upper-cased and used as is (allsteps.py:84-125)."""
    if isinstance(telopattern, list):
        return [p.upper() for p in telopattern]
    if "|" in telopattern:
        # upstream returns a *string* here that its callers then iterate character by character
        # (allsteps.py:90-102, 168); README.md:205 calls the form untested.  Not supported.
        raise ValueError("'|' alternation patterns are not supported")
    fwd = pattern_scramble_telo(telopattern, [cut_length])
    return [p.upper() for p in fwd + [p.translate(_COMPLEMENT) for p in fwd]]


def unzip_file(filepath):
    """Records of a FASTA/FASTQ(.gz) file (allsteps.py:127-149)."""
    return seqio.read_records(filepath)


def min_count_for_cutoff(cutoff: float, ratio: float, no_bp: int) -> int:
    """Largest integer count c for which `c / ratio > cutoff` is False in float64, so that the
    kernel's integer test `count > min_count` equals the reference's float test exactly."""
    c = int(np.floor(cutoff * ratio)) if np.isfinite(cutoff * ratio) else -1
    c = max(min(c, no_bp + 1), -1)
    while c >= 0 and (c / ratio > cutoff):
        c -= 1
    while (c + 1) / ratio <= cutoff and c + 1 <= no_bp + 1:
        c += 1
    return c


# ---------------------------------------------------------------------------- a3
def patternTRC_count(filepath, telopattern, read_length=0, kmer=4, no_bp=1000, cutoff=0.5):
    """Step 1 (allsteps.py:152-204): [[id, best_pattern, 'forward'|'reverse', trc], ...] for reads
    longer than read_length whose TRC exceeds cutoff."""
    if isinstance(filepath, list):
        print("Can only process 1 file path at the time, please loop paths through the list")
        return None
    from . import batch
    patterns = patterns_to_search(telopattern, cut_length=kmer)
    ratio = no_bp / len(telopattern)
    eng = get_engine()
    eng.set_patterns(patterns)
    rows = []
    prm = hiplib.make_params(no_bp=no_bp, min_len=read_length, min_count=min_count_for_cutoff(cutoff, ratio, no_bp),
                             flags=hiplib.F_STEP1)
    for recs in batch.record_batches(seqio.read_records(filepath)):
        bases, offsets = hiplib.pack_reads([r.seq for r in recs])
        eng.upload(0, bases, offsets)
        eng.scan(0, prm)
        eng.sync()
        res = eng.results(0)
        for r, rec in zip(res, recs):
            if r["pass"]:
                fwd = r["tail"] == 0
                cnt, idx = (r["best_start"], r["best_start_idx"]) if fwd else (r["best_end"], r["best_end_idx"])
                rows.append([rec.id, patterns[idx], "forward" if fwd else "reverse", int(cnt) / ratio])
    if check_file_type(filepath) is None:
        print("can not read in file - can not run step 1")
        return None
    return rows


# ---------------------------------------------------------------------------- a4
def seq_cut_windows(s, window_size, step):
    """(start, text) of every window; the text is window_size-1 characters long, which is the
    reference's behaviour (allsteps.py:219-224)."""
    return [(a, s[a:a + window_size - 1]) for a in range(0, len(s) - window_size + 1, step)]


# ---------------------------------------------------------------------------- a5 / a6
def _window_scan(seq, tails, patterns, windowSize, slide, trimfirst, maxlengthtelo, raw=False):
    eng = get_engine()
    eng.set_patterns(patterns)
    bases, offsets = hiplib.pack_reads([seq] * len(tails))
    tv = np.array([0 if t == "forward" else 1 for t in tails], dtype=np.uint8)
    return eng.window_counts(bases, offsets, tv, windowSize, slide, trimfirst, maxlengthtelo, raw=raw)


def bound_detect(filepath, read, pattern_telo, windowSize, slide, trimfirst, maxlengthtelo, cut_length,
                 tail=None, plot_yes_no=None, plotcp_range=None):
    """Step 2 for one read (allsteps.py:227-338): [[read, boundary_bp]] for the requested tail
    (both tails, reverse first, when tail is None)."""
    if not isinstance(read, str):
        print("can only read in 1 read at a time")
        return None
    patterns = patterns_to_search(pattern_telo, cut_length=cut_length)
    boundary = []
    for rec in seqio.read_records(filepath):
        if rec.id != read or windowSize is None:
            continue
        # upstream OVERWRITES maxlengthtelo with a shorter record's length (allsteps.py:263-264) and has no `break`: with duplicate
        # read ids a later record of the same id is clipped by the shortest one before it as well (the only place
        # the per-read API and the reference could be made to disagree)
        m = maxlengthtelo = min(maxlengthtelo, len(rec.seq))
        tails = [t for t in ("reverse", "forward") if tail in (None, t)]
        sums, win_off, _ = _window_scan(rec.seq, tails, patterns, windowSize, slide, trimfirst, m)
        eng = get_engine()
        for i, t in enumerate(tails):
            s_w = sums[win_off[i]:win_off[i + 1]]
            if len(s_w) == 0:
                continue
            bkp, _ = eng.binseg_l2(s_w, np.array([0, len(s_w)], dtype=np.int64), len(patterns))
            if bkp[0] < 0:
                raise BadSegmentationParameters(f"{len(s_w)} windows for read {read}")
            point = int(bkp[0]) * slide + trimfirst
            if plot_yes_no:
                _plot_changepoint(rec.id, s_w / len(patterns), slide, trimfirst, point, plotcp_range or m)
            boundary.append([read, point if (point <= m and point != 0) else 0])
    return boundary


def _plot_changepoint(read_id, y, slide, trimfirst, point, xlim):
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    x = np.arange(len(y)) * slide + trimfirst
    plt.figure(figsize=(7.5, 3), dpi=300)
    plt.plot(x, y, color="#000000", linestyle="-", linewidth=2)
    plt.axvline(x=point, color="#FF2C2C", linewidth=2, linestyle="--", label=f"x = boundary point: {point}")
    plt.title(f"mean window + boundary point of {read_id}")
    plt.xlabel("base pair (bp)")
    plt.ylabel("mean window value")
    plt.xlim(0, xlim)
    plt.tight_layout()
    plt.grid(True)


# ---------------------------------------------------------------------------- a7
def rawCountPattern(filepath, read, pattern_telo, windowSize, slide, trimfirst, cut_length, minSeqLength,
                    maxlengthtelo, tail=None, plot_raw=False):
    """Per-window, per-pattern counts (`matches or 1`) as a DataFrame with the reference's columns
    ['tail', 'position', 'pattern', 'count'], window-major then pattern order, forward rows before
    reverse rows (allsteps.py:359-419, 464)."""
    import pandas as pd
    if not isinstance(read, str):
        print("can only read in 1 read at a time")
        return None
    patterns = patterns_to_search(pattern_telo, cut_length=cut_length)
    frames = []
    for rec in seqio.read_records(filepath):
        if rec.id != read or windowSize is None:
            continue
        tails = [t for t in ("forward", "reverse") if tail in (None, t)]
        _, win_off, raw = _window_scan(rec.seq, tails, patterns, windowSize, slide, trimfirst,
                                       min(maxlengthtelo, len(rec.seq)), raw=True)
        for i, t in enumerate(tails):
            block = raw[win_off[i]:win_off[i + 1]]
            n_win = block.shape[0]
            frames.append(pd.DataFrame({
                "tail": t,
                "position": np.repeat(np.arange(n_win, dtype=np.int64) * slide, len(patterns)),
                "pattern": np.tile(np.array(patterns, dtype=object), n_win),
                "count": block.reshape(-1).astype(np.int64),
            }))
    if not frames:
        return pd.DataFrame([], columns=["tail", "position", "pattern", "count"])
    return pd.concat(frames, ignore_index=True)[["tail", "position", "pattern", "count"]]


# ---------------------------------------------------------------------------- f3
def fit_quadratic_and_find_vertex(trc_list, telo_length_list, inputtrc, median_trc, save_path=None):
    """Degree-2 fit of telomere length on TRC and its clamped vertex (allsteps.py:467-501)."""
    trc = np.array(trc_list)
    telo = np.array(telo_length_list)
    coeffs = np.polyfit(trc, telo, 2)
    a, b, c = coeffs
    vertex_x = -b / (2 * a)
    if vertex_x > 1.0:
        vertex_x = median_trc
    if vertex_x < inputtrc:
        vertex_x = inputtrc
    vertex_y = a * vertex_x ** 2 + b * vertex_x + c
    if save_path:
        plot_quadratic_fit(trc, telo, coeffs, vertex_x, vertex_y, save_path)
    return vertex_x, vertex_y, coeffs


def plot_quadratic_fit(trc, telo, coeffs, vertex_x, vertex_y, save_path):
    """The PNG of allsteps.py:484-499.  Object-oriented matplotlib (no pyplot state): safe to run on a helper thread, which is
    where the CLI runs it -- the fit's numbers are logged at once, the picture follows (main.summarize)."""
    from matplotlib.backends.backend_agg import FigureCanvasAgg
    from matplotlib.figure import Figure
    a, b, c = coeffs
    xs = np.linspace(min(trc), max(trc), 100)
    fig = Figure(figsize=(7, 5))
    FigureCanvasAgg(fig)
    ax = fig.add_subplot(111)
    ax.scatter(trc, telo, color="blue", label="Topsicle results")
    ax.plot(xs, a * xs ** 2 + b * xs + c, color="red", label="Fit line")
    ax.scatter([vertex_x], [vertex_y], color="green", label="Vertex")
    ax.set_xlabel("TRC values")
    ax.set_ylabel("Telomere length, each read (bp)")
    ax.set_title("Quadratic fit plot")
    ax.legend()
    fig.tight_layout()
    fig.savefig(save_path, dpi=300)
