"""The exploratory plots (Topsicle/descriptive_plot.py, overview_plot.py upstream): the rows behind the
k-mer / following-bases heatmap and the motif positions must equal what the reference's own code produced on
the demo file (tests/golden/demo_overview.json, written by oracle/gen_golden.py in the build container)."""
import json
import os

import pandas as pd
import pytest

from topsicle_amd import descriptive_plot as dp
from topsicle_amd import seqio


@pytest.fixture(scope="module")
def gold(gold_dir):
    return json.load(open(os.path.join(gold_dir, "demo_overview.json")))


def test_heatmap_rows_equal_reference(gold, gold_dir):
    demo = os.path.join(gold_dir, "demo_col0.fastq.gz")
    recs = list(seqio.read_records(demo))
    for h in gold["heatmaps"]:
        pats, rows = dp.pattern_matches(recs, h["motif"], h["k"], h["minSeqLength"])
        assert len(rows) == h["n_rows"]
        assert [[r[0], r[1], r[2]] for r in rows[:25]] == h["first_rows"]
        df = pd.DataFrame(rows, columns=["Pattern", "Match", "read id"])
        tab = pd.crosstab(df["Match"], df["Pattern"])
        assert [str(c) for c in tab.columns] == h["patterns"] and [str(i) for i in tab.index] == h["matches"]
        assert tab.values.astype(int).tolist() == h["counts"]


def test_descriptive_positions_equal_reference(gold, gold_dir):
    recs = {r.id: r for r in seqio.read_records(os.path.join(gold_dir, "demo_col0.fastq.gz"))}
    for g in gold["positions"]:
        got = dp.match_positions(recs[g["id"]].seq, g["motif"], g["minSeqLength"])
        assert {k: [list(v[0]), list(v[1])] for k, v in got.items()} == g["pos"]


def test_plots_render(tmp_path, gold_dir):
    """The drawing code runs headless and returns what upstream returns."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    demo = os.path.join(gold_dir, "demo_col0.fastq.gz")
    assert dp.descriptive_plot(demo, "CCCTAAA", 9000) == "plotted"
    plt.savefig(tmp_path / "d.png", dpi=50)
    plt.close("all")
    df = dp.patterns_vs_match_heatmap(demo, "CCCTAAA", 5, 9000)
    plt.savefig(tmp_path / "h.png", dpi=50)
    plt.close("all")
    assert list(df.columns) == ["Pattern", "Match", "read id"] and len(df) == 24200
    assert os.path.getsize(tmp_path / "d.png") > 1000 and os.path.getsize(tmp_path / "h.png") > 1000
