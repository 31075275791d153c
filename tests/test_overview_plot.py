"""The k-mer / following-bases counts behind the overview heat map (Topsicle/descriptive_plot.py:259-291):
the kernel's picks and histogram must give exactly the rows and the crosstab the reference's own code produced on the demo
file (tests/golden/demo_overview.json, written by oracle/gen_golden.py from the imported reference), and the oracle's on
random reads.  CPU: the kernel source through the emulation; -m gpu: the real kernel through the C ABI."""
import json
import os

import numpy as np
import pandas as pd
import pytest

import topsicle_oracle as orc
from topsicle_amd import allsteps, descriptive_plot as dp, seqio


@pytest.fixture(scope="module")
def gold(gold_dir):
    return json.load(open(os.path.join(gold_dir, "demo_overview.json")))


@pytest.fixture()
def emu_engine():
    from emu_engine import EmuEngine
    e = EmuEngine()
    allsteps.set_engine(e)
    yield e
    allsteps.set_engine(None)


@pytest.fixture()
def gpu_engine():
    from topsicle_amd import hiplib
    e = hiplib.HipScanner(0)
    allsteps.set_engine(e)
    yield e
    allsteps.set_engine(None)
    e.close()


def _check_heatmaps_against_reference(gold, gold_dir, engine):
    recs = list(seqio.read_records(os.path.join(gold_dir, "demo_col0.fastq.gz")))
    for h in gold["heatmaps"]:
        pats, rows, counts = dp.pattern_matches(recs, h["motif"], h["k"], h["minSeqLength"], engine)
        assert len(rows) == h["n_rows"]
        assert [[r[0], r[1], r[2]] for r in rows[:25]] == h["first_rows"]
        df = pd.DataFrame(rows, columns=["Pattern", "Match", "read id"])
        tab = pd.crosstab(df["Match"], df["Pattern"])
        assert [str(c) for c in tab.columns] == h["patterns"] and [str(i) for i in tab.index] == h["matches"]
        assert tab.values.astype(int).tolist() == h["counts"]
        # the device-side crosstab: same numbers, bins in 2-bit code order (+ one bin for non-ACGT followers)
        follow = len(h["motif"]) - h["k"]
        labels = dp.follower_labels(follow)
        assert counts.shape == (len(pats), 4 ** follow + 1) and int(counts.sum()) == h["n_rows"]
        for j, p in enumerate(pats):
            for b, lab in enumerate(labels):
                want = h["counts"][h["matches"].index(lab)][h["patterns"].index(p)] if lab in h["matches"] and p in h["patterns"] else 0
                assert counts[j, b] == want, (p, lab)


def _check_random_reads_against_oracle(engine, seed):
    rng = np.random.default_rng(seed)
    motif, k = [("CCCTAA", 4), ("CCCTAAA", 5), ("TTAGGG", 3), ("AAACCCT", 7), ("CCCTAA", 6)][seed % 5]
    seqs = []
    for L in [0, 99, 104, 150, 1999, 2000, 2001, 2500, 5000, 9000]:
        tract = int(rng.integers(0, max(1, L)))
        ph = int(rng.integers(len(motif)))
        body = list(((motif * (tract // len(motif) + 2))[ph:ph + tract] + "".join("ACGT"[x] for x in rng.integers(0, 4, max(0, L - tract))))[:L])
        for p in rng.integers(0, max(1, L), L // 30):
            if body:
                body[p] = "ACGTNacgtnR"[int(rng.integers(11))]
        s = "".join(body)
        seqs.append(s if rng.random() < 0.5 else s[::-1].translate(str.maketrans("ACGTacgt", "TGCAtgca")))
    recs = [seqio.Record(f"r{i}", f"r{i}", s) for i, s in enumerate(seqs)]
    pats, rows, counts = dp.pattern_matches(recs, motif, k, 120, engine)
    want = []
    for strand in (0, 1):
        for r in recs:
            if len(r.seq) > 120:
                want += [(p, m, [r.id]) for p, m, _pos in orc.kmer_followers(r.seq, motif, k)[strand]]
    assert rows == want
    follow = len(motif) - k
    labels = dp.follower_labels(follow)
    for j, p in enumerate(pats):
        mine = [m for q, m, _ in want if q == p]
        for b, lab in enumerate(labels):
            assert counts[j, b] == mine.count(lab)
        assert counts[j, -1] == sum(1 for m in mine if set(m) - set("ACGT"))


def test_heatmap_rows_equal_reference_emulation(gold, gold_dir, emu_engine):
    _check_heatmaps_against_reference(gold, gold_dir, emu_engine)


@pytest.mark.parametrize("seed", range(5))
def test_followers_random_vs_oracle_emulation(emu_engine, seed):
    _check_random_reads_against_oracle(emu_engine, seed)


def test_descriptive_positions_equal_reference(gold, gold_dir):
    recs = {r.id: r for r in seqio.read_records(os.path.join(gold_dir, "demo_col0.fastq.gz"))}
    for g in gold["positions"]:
        got = dp.match_positions(recs[g["id"]].seq, g["motif"], g["minSeqLength"])
        assert {k: [list(v[0]), list(v[1])] for k, v in got.items()} == g["pos"]


def test_overview_driver_with_emulation(tmp_path, gold_dir, emu_engine):
    """overview_plot end to end: TRC filter (kernel step 1), follower counts (kernel), plots and the raw-count CSV."""
    from topsicle_amd import overview_plot
    out = tmp_path / "ov"
    args = overview_plot.build_parser().parse_args(["--inputDir", os.path.join(gold_dir, "demo_col0.fastq.gz"), "--outputDir", str(out),
                                                    "--pattern", "CCCTAAA", "--recfindingpattern", "--rawcount"])
    overview_plot.run(args, engines=[emu_engine])
    assert (out / "descriptive_plot_1.png").stat().st_size > 1000 and (out / "heatmap_1.png").stat().st_size > 1000
    df = pd.read_csv(out / "heatmap_rawcount_1.csv")
    assert list(df.columns) == ["Pattern", "Match", "read id"]
    gold_ids = {r.split(",")[3] for r in open(os.path.join(gold_dir, "demo_telolengths_all.csv")).read().splitlines()[1:]}
    assert {x.strip("[]'") for x in df["read id"].unique()} == gold_ids and len(df) > 5000


@pytest.mark.gpu
def test_heatmap_rows_equal_reference_on_gpu(gold, gold_dir, gpu_engine):
    _check_heatmaps_against_reference(gold, gold_dir, gpu_engine)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(5))
def test_followers_random_vs_oracle_on_gpu(gpu_engine, seed):
    _check_random_reads_against_oracle(gpu_engine, seed)


@pytest.mark.gpu
def test_followers_batch_properties_on_gpu(gpu_engine):
    """4000 config-shaped reads: per-read picks are non-overlapping by construction, the histogram equals the picks'
    count, and the reverse-complemented batch swaps the two strands."""
    from topsicle_amd import hiplib, synth
    motif, k = "CCCTAA", 4
    pats = allsteps.patterns_to_search(motif, k)
    gpu_engine.set_patterns(pats)
    n, L = 4000, 15000
    bases, offsets, _ = synth.make_reads(n, L, motif, seed=5)
    gpu_engine.upload(1, bases, offsets)
    picks, hist = gpu_engine.kmer_followers(1, 6, 2, 100, 2000, 9000)
    assert int(np.unpackbits(picks.view(np.uint8)).sum()) == int(hist.sum()) > n * 100
    comp = np.zeros(256, np.uint8)
    comp[list(b"ACGT")] = list(b"TGCA")
    rc = comp[bases.reshape(n, L)[:, ::-1]].reshape(-1)
    gpu_engine.upload(2, rc, offsets)
    picks_rc, hist_rc = gpu_engine.kmer_followers(2, 6, 2, 100, 2000, 9000)
    assert np.array_equal(picks[:, 0], picks_rc[:, 1]) and np.array_equal(picks[:, 1], picks_rc[:, 0])
    assert np.array_equal(hist[0], hist_rc[1]) and np.array_equal(hist[1], hist_rc[0])
    for i in (0, 77, 3999):
        seq = bytes(bases[offsets[i]:offsets[i + 1]]).decode()
        for strand, rows in enumerate(orc.kmer_followers(seq, motif, k)):
            for j, p in enumerate(orc.kmers_of_repeat(motif, k)):
                bits = np.unpackbits(picks[i, strand, j].view(np.uint8), bitorder="little")
                assert np.flatnonzero(bits).tolist() == [pos for q, _m, pos in rows if q == p]


@pytest.mark.gpu
def test_overview_driver_on_gpu(tmp_path, gold_dir):
    from topsicle_amd import overview_plot
    out = tmp_path / "ov"
    overview_plot.main(["--inputDir", os.path.join(gold_dir, "demo_col0.fastq.gz"), "--outputDir", str(out),
                        "--pattern", "CCCTAAA", "--recfindingpattern", "--rawcount"])
    assert (out / "heatmap_1.png").exists()
    assert (out / "descriptive_plot_1.png").stat().st_size > 1000        # upstream always writes the scatter (overview_plot.py:92)
    df = pd.read_csv(out / "heatmap_rawcount_1.csv")
    assert list(df.columns) == ["Pattern", "Match", "read id"] and len(df) > 5000
