"""Host-side mirror of Topsicle/descriptive_plot.py (exploratory plots; reference file:line cited per function).

These plots run on the FEW reads that passed the TRC filter (the filter itself is the GPU step 1,
`allsteps.patternTRC_count`), over at most 9 kb / 1.9 kb per read, so the counting stays on the host:
literal `re.finditer` searches like upstream, which also makes the result identical by construction.
Drawing uses matplotlib only (the reference styles with seaborn; the numbers are the same).
"""
from __future__ import annotations

import re

from . import seqio
from .allsteps import pattern_scramble_telo

_COMPLEMENT = str.maketrans("ACGT", "TGCA")


def match_positions(seq: str, pattern: str, minSeqLength: int):
    """Start positions of the (non-overlapping) occurrences of the motif and of its complement in the
    first `minSeqLength` bases of the read and of the reversed read (descriptive_plot.py:103-136).
    Returns {pattern: (positions in seq, positions in reversed seq)} for the two patterns."""
    pats = [pattern.upper(), pattern.translate(_COMPLEMENT).upper()]
    s1 = seq[:minSeqLength].upper()
    s2 = seq[::-1][:minSeqLength].upper()
    return {p: ([m.start() for m in re.finditer(re.escape(p), s1)], [m.start() for m in re.finditer(re.escape(p), s2)])
            for p in pats}


def descriptive_plot(filepath, pattern, minSeqLength):
    """Location of the telomere motif (and its complement) along the first `minSeqLength` bases of
    both ends of every read longer than `minSeqLength`, at most 41 reads (descriptive_plot.py:89-165)."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt

    if seqio.check_file_type(filepath) is None:
        print("problem in filepath, can not have descriptive plot")
        return None
    file_name = filepath.split("/")[-1].split(".")[0]
    fig, ax = plt.subplots(figsize=(10, 15))
    colors = ["#0173b2", "#de8f05"]
    labels = [f"5'-{pattern.upper()}-3'", f"3'-{pattern.translate(_COMPLEMENT).upper()}-5'"]
    read_ids = []
    k_line = 0
    for rec in seqio.read_records(filepath):
        if len(rec.seq) <= minSeqLength:
            continue
        read_ids.append(rec.id)
        for i, (pat, (m1, m2)) in enumerate(match_positions(rec.seq, pattern, minSeqLength).items()):
            ax.scatter(m1 + m2, [k_line] * (len(m1) + len(m2)), color=colors[i], marker="|",
                       label=labels[i] if k_line == 0 else None, zorder=2)
        k_line += 2
        if len(read_ids) > 40:
            print("file has more than 40 reads, but it is not recommended to have plot with that many reads")
            print("so the output plot will have 40 reads only")
            break
    ax.set_title(f"Location of telomere patterns in {file_name}")
    ax.set_xlabel("Position")
    if read_ids:
        ax.legend(title="Pattern")
    ax.set_yticks([i * 2 for i in range(len(read_ids))])
    ax.set_yticklabels(read_ids)
    ax.grid(True, color="grey", linestyle="--")
    plt.tight_layout()
    return "plotted"


def pattern_matches(records, telopattern, telophrase, minSeqLength):
    """The rows behind the heatmap (descriptive_plot.py:259-291): for every read longer than `minSeqLength`,
    in bases 100..2000 of the read and of its reverse complement, every non-overlapping occurrence of each
    k-mer of the doubled motif followed by `len(motif) - k` more bases -> (k-mer, those bases, [read id])."""
    pattern_all = pattern_scramble_telo(telopattern, cut_length=telophrase)
    finding = int(len(telopattern) - telophrase)
    regexes = [(p, re.compile(rf"{re.escape(p)}(.{{{finding}}})")) for p in pattern_all]
    rows_1, rows_2 = [], []
    for rec in records:
        if len(rec.seq) <= minSeqLength:
            continue
        seq = rec.seq[100:2000].upper()
        seq_2 = rec.seq[::-1][100:2000].upper().translate(_COMPLEMENT)
        for p, rx in regexes:
            rows_1 += [(p, m.group(1), [rec.id]) for m in rx.finditer(seq)]
            rows_2 += [(p, m.group(1), [rec.id]) for m in rx.finditer(seq_2)]
    return pattern_all, rows_1 + rows_2


def patterns_vs_match_heatmap(filepath, telopattern, telophrase, minSeqLength):
    """Heatmap of k-mer vs following bases; returns the DataFrame of all matches with the columns
    ["Pattern", "Match", "read id"] like upstream (descriptive_plot.py:233-313)."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    import pandas as pd

    if seqio.check_file_type(filepath) is None:
        print("problem in filepath, can not have heatmap")
        return None
    file_name = filepath.split("/")[-1].split(".")[0]
    pattern_all, rows = pattern_matches(seqio.read_records(filepath), telopattern, telophrase, minSeqLength)
    print(pattern_all)
    allstrands = pd.DataFrame(rows, columns=["Pattern", "Match", "read id"])
    match_order = sorted(allstrands["Match"].dropna().unique())
    allstrands["Match"] = pd.Categorical(allstrands["Match"], categories=match_order, ordered=True)

    hist_data = pd.crosstab(allstrands["Match"], allstrands["Pattern"])
    fig, ax = plt.subplots(figsize=(8, 8), dpi=300)
    if hist_data.size:
        im = ax.imshow(hist_data.values, cmap="Blues", aspect="auto")
        fig.colorbar(im, ax=ax, shrink=0.75)
        for (i, j), v in _ndenumerate(hist_data.values):
            ax.text(j, i, str(int(v)), ha="center", va="center", fontsize=6)
        ax.set_xticks(range(hist_data.shape[1]))
        ax.set_xticklabels(hist_data.columns, rotation=45, ha="right")
        ax.set_yticks(range(hist_data.shape[0]))
        ax.set_yticklabels(hist_data.index)
    ax.set_ylabel("Match")
    ax.set_xlabel("Pattern")
    plt.suptitle(f"{telophrase}-bp patterns and matches from reads in \n {file_name}")
    plt.tight_layout()
    return allstrands


def _ndenumerate(a):
    for i in range(a.shape[0]):
        for j in range(a.shape[1]):
            yield (i, j), a[i, j]
