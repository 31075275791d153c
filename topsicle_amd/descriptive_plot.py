"""Topsicle/descriptive_plot.py's two exploratory plots (reference file:line cited per function).

The heat map's counting -- every k-mer of the doubled motif followed by the next len(motif) - k bases, bases 100..2000 of
both strands (descriptive_plot.py:259-291) -- runs on the GPU (tps_batch_kmer_followers): the kernel
returns one bit per match position, from which the reference's DataFrame rows are built, and the crosstab itself.  Drawing
uses matplotlib only.  The scatter of whole-motif hits (descriptive_plot.py:89-165: visualisation of at most 41 reads, no part
of the hot path) is a host-side literal search over the few reads that passed the TRC filter on the GPU.
"""
from __future__ import annotations

import re

from . import seqio
from .allsteps import pattern_scramble_telo

_COMPLEMENT = str.maketrans("ACGT", "TGCA")
LO, HI = 100, 2000               # the stretch of either end the heat map looks at (descriptive_plot.py:264-266)
MAX_READS_DRAWN = 41             # the scatter stops after this many reads (descriptive_plot.py:146-150)


def _file_label(filepath: str) -> str:
    return filepath.split("/")[-1].split(".")[0]


# ---------------------------------------------------------------------------- scatter of whole-motif hits
def match_positions(seq: str, pattern: str, minSeqLength: int):
    """Start positions of the (non-overlapping) occurrences of the motif and of its complement in the
    first `minSeqLength` bases of the read and of the reversed read (descriptive_plot.py:103-136).
    Returns {pattern: (positions in seq, positions in reversed seq)} for the two patterns."""
    pats = [pattern.upper(), pattern.translate(_COMPLEMENT).upper()]
    s1 = seq[:minSeqLength].upper()
    s2 = seq[::-1][:minSeqLength].upper()
    return {p: ([m.start() for m in re.finditer(re.escape(p), s1)], [m.start() for m in re.finditer(re.escape(p), s2)])
            for p in pats}


def descriptive_plot_records(records, label, pattern, minSeqLength):
    """The scatter for records already in memory; returns the ids drawn."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    _fig, ax = plt.subplots(figsize=(10, 15))
    colors = ["#0173b2", "#de8f05"]
    labels = [f"5'-{pattern.upper()}-3'", f"3'-{pattern.translate(_COMPLEMENT).upper()}-5'"]
    drawn = []
    for rec in records:
        if len(rec.seq) <= minSeqLength:
            continue
        y = 2 * len(drawn)
        for i, (m1, m2) in enumerate(match_positions(rec.seq, pattern, minSeqLength).values()):
            ax.scatter(m1 + m2, [y] * (len(m1) + len(m2)), color=colors[i], marker="|", label=None if drawn else labels[i], zorder=2)
        drawn.append(rec.id)
        if len(drawn) >= MAX_READS_DRAWN:
            print("file has more than 40 reads, but it is not recommended to have plot with that many reads")
            print("so the output plot will have 40 reads only")
            break
    ax.set_title(f"Location of telomere patterns in {label}")
    ax.set_xlabel("Position")
    if drawn:
        ax.legend(title="Pattern")
    ax.set_yticks([2 * i for i in range(len(drawn))])
    ax.set_yticklabels(drawn)
    ax.grid(True, color="grey", linestyle="--")
    plt.tight_layout()
    return drawn


def descriptive_plot(filepath, pattern, minSeqLength):
    """Location of the telomere motif (and its complement) along the first `minSeqLength` bases of
    both ends of every read longer than `minSeqLength`, at most 41 reads (descriptive_plot.py:89-165)."""
    if seqio.check_file_type(filepath) is None:
        print("problem in filepath, can not have descriptive plot")
        return None
    descriptive_plot_records(seqio.read_records(filepath), _file_label(filepath), pattern, minSeqLength)
    return "plotted"


# ---------------------------------------------------------------------------- k-mer followers (the heat map)
def follower_rows(seqs, ids, picks, pattern_all, follow):
    """The reference's DataFrame rows from the kernel's pick bits: all rows of the reads themselves (read by read, k-mer
    by k-mer, left to right), then all rows of the reverse complements (descriptive_plot.py:280-295).  `picks` is
    uint32[n, 2, n_fwd, pw] from HipScanner.kmer_followers; the following letters are read from the host's copy."""
    import numpy as np
    k = len(pattern_all[0])
    rows = []
    for strand in (0, 1):
        for i, seq in enumerate(seqs):
            if not picks[i, strand].any():
                continue
            s = seq[LO:HI].upper() if strand == 0 else seq[::-1][LO:HI].upper().translate(_COMPLEMENT)
            for j, p in enumerate(pattern_all):
                bits = np.unpackbits(picks[i, strand, j].view(np.uint8), bitorder="little")
                rows += [(p, s[pos + k:pos + k + follow], [ids[i]]) for pos in np.flatnonzero(bits).tolist()]
    return rows


def pattern_matches(records, telopattern, telophrase, minSeqLength, engine=None):
    """The rows behind the heatmap (descriptive_plot.py:259-291): for every read longer than `minSeqLength`, in bases
    100..2000 of the read and of its reverse complement, every non-overlapping occurrence of each k-mer of the doubled
    motif followed by `len(motif) - k` more bases -> (k-mer, those bases, [read id]).  The matching runs on the GPU
    (tps_batch_kmer_followers); returns (k-mers, rows, counts int64[n_kmers, 4**follow + 1] summed over both strands)."""
    from . import allsteps, hiplib
    pattern_all = pattern_scramble_telo(telopattern, cut_length=telophrase)
    follow = int(len(telopattern) - telophrase)
    eng = engine or allsteps.get_engine()
    eng.set_patterns(allsteps.patterns_to_search(telopattern, telophrase))
    recs = [r for r in records if len(r.seq) > minSeqLength]
    if not recs:
        return pattern_all, [], None
    bases, offsets = hiplib.pack_reads([r.seq for r in recs])
    eng.upload(0, bases, offsets)
    picks, hist = eng.kmer_followers(0, len(pattern_all), follow, LO, HI, minSeqLength)
    rows = follower_rows([r.seq for r in recs], [r.id for r in recs], picks, pattern_all, follow)
    return pattern_all, rows, hist.sum(axis=0)


def follower_labels(follow):
    """Match strings of the device histogram's bins, in bin order (2-bit codes A C T G = 0 1 2 3, first base lowest)."""
    return ["".join("ACTG"[(c >> (2 * j)) & 3] for j in range(follow)) for c in range(4 ** follow)]


def heatmap_from_records(records, label, telopattern, telophrase, minSeqLength, engine=None):
    """Draws the heat map for records in memory; returns the DataFrame of all matches, columns
    ["Pattern", "Match", "read id"] like upstream (descriptive_plot.py:293-313)."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    import pandas as pd
    pattern_all, rows, _counts = pattern_matches(records, telopattern, telophrase, minSeqLength, engine)
    print(pattern_all)
    table = pd.DataFrame(rows, columns=["Pattern", "Match", "read id"])
    order = sorted(table["Match"].dropna().unique())
    table["Match"] = pd.Categorical(table["Match"], categories=order, ordered=True)
    grid = pd.crosstab(table["Match"], table["Pattern"])
    fig, ax = plt.subplots(figsize=(8, 8), dpi=300)
    if grid.size:
        im = ax.imshow(grid.values, cmap="Blues", aspect="auto")
        fig.colorbar(im, ax=ax, shrink=0.75)
        for i in range(grid.shape[0]):
            for j in range(grid.shape[1]):
                ax.text(j, i, str(int(grid.values[i, j])), ha="center", va="center", fontsize=6)
        ax.set_xticks(range(grid.shape[1]))
        ax.set_xticklabels(grid.columns, rotation=45, ha="right")
        ax.set_yticks(range(grid.shape[0]))
        ax.set_yticklabels(grid.index)
    ax.set_ylabel("Match")
    ax.set_xlabel("Pattern")
    plt.suptitle(f"{telophrase}-bp patterns and matches from reads in \n {label}")
    plt.tight_layout()
    return table


def patterns_vs_match_heatmap(filepath, telopattern, telophrase, minSeqLength):
    """Heatmap of k-mer vs following bases for the reads of a file (descriptive_plot.py:233-313)."""
    if seqio.check_file_type(filepath) is None:
        print("problem in filepath, can not have heatmap")
        return None
    return heatmap_from_records(list(seqio.read_records(filepath)), _file_label(filepath), telopattern, telophrase, minSeqLength)
