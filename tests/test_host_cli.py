"""Host-side logic without a GPU: the allsteps mirror, the batched driver and the `topsicle`
CLI run on the demo data with the emulated engine and must reproduce the reference's shipped
results (Topsicle_demo/telolengths_all.csv, log lines) and the reference-generated goldens."""
import csv
import gzip
import json
import os
import shutil

import numpy as np
import pytest

from emu_engine import EmuEngine
from topsicle_amd import allsteps, main as cli, seqio


@pytest.fixture()
def engine():
    e = EmuEngine()
    allsteps.set_engine(e)
    yield e
    allsteps.set_engine(None)


@pytest.fixture()
def demo_fastq(tmp_path, gold_dir):
    d = tmp_path / "in"
    d.mkdir()
    dst = d / "Col-0-6909_GWHBDNP00000001.1_nano_right.fastq.gz"
    shutil.copyfile(os.path.join(gold_dir, "demo_col0.fastq.gz"), dst)
    return str(dst)


def run_cli(engine, argv):
    args = cli.build_parser().parse_args(argv)
    cli.tprint.logfile = cli.get_log_path(args)
    cli.analysis_run(args, engines=[engine])
    return args


def test_pattern_functions(gold_dir):
    for c in json.load(open(os.path.join(gold_dir, "patterns.json"))):
        if c["scramble"] is not None:
            assert allsteps.pattern_scramble_telo(c["motif"], c["k"]) == c["scramble"]
        assert allsteps.patterns_to_search(c["motif"], c["k"]) == c["search"]


def test_seq_cut_windows():
    w = allsteps.seq_cut_windows("ACGTACGTACGT", 5, 3)
    assert w == [(0, "ACGT"), (3, "TACG"), (6, "GTAC")]


def test_min_count_for_cutoff_matches_float_test():
    for motif_len in (5, 6, 7, 12):
        ratio = 1000 / motif_len
        for cutoff in (-1.0, 0.0, 0.3, 0.5, 0.7, 0.7000000001, 0.8, 1.0, 5.0):
            mc = allsteps.min_count_for_cutoff(cutoff, ratio, 1000)
            for c in range(0, 1002):
                assert (c > mc) == (c / ratio > cutoff), (motif_len, cutoff, c)


def test_patternTRC_count_demo(engine, demo_fastq, gold_dir):
    for case in json.load(open(os.path.join(gold_dir, "demo_step1.json"))):
        rows = allsteps.patternTRC_count(demo_fastq, case["motif"], read_length=case["min_len"], kmer=case["k"],
                                         no_bp=1000, cutoff=case["cutoff"])
        assert rows == case["rows"], (case["motif"], case["k"], case["cutoff"])


def test_bound_detect_and_rawcount_demo(engine, demo_fastq, demo_windows):
    meta, arrs = demo_windows
    pats = meta["patterns"]
    for i, r in enumerate(meta["reads"][:17:4]):
        i = i * 4
        b = allsteps.bound_detect(demo_fastq, r["id"], pats, 100, 6, 100, 20000, 5, tail=r["tail"])
        assert b == [[r["id"], r["boundary"]]]
        df = allsteps.rawCountPattern(demo_fastq, r["id"], pats, 100, 6, 100, 5, 9000, 20000, tail=r["tail"])
        want = arrs[f"counts_{i}"]
        assert list(df.columns) == ["tail", "position", "pattern", "count"]
        assert np.array_equal(df["count"].to_numpy().reshape(-1, len(pats)), want)
        assert df["pattern"].tolist()[: len(pats)] == pats and set(df["tail"]) == {r["tail"]}
        assert df["position"].tolist()[len(pats)] == 6
    # tail=None: reverse boundary first, then forward (allsteps.py:335-336)
    rid = meta["reads"][0]["id"]
    both = allsteps.bound_detect(demo_fastq, rid, pats, 100, 7, 100, 20000, 5)
    s7 = {m["tail"]: m["boundary"] for m in meta["reads"] if m.get("key", "").startswith("s7_0")}
    assert both == [[rid, s7["reverse"]], [rid, s7["forward"]]]


def test_cli_demo_reproduces_reference_outputs(engine, demo_fastq, tmp_path, gold_dir, capsys):
    out = tmp_path / "out"
    run_cli(engine, ["--inputDir", os.path.dirname(demo_fastq), "--outputDir", str(out), "--pattern", "CCCTAAA", "--slide", "6"])
    got = open(out / "telolengths_all.csv").read().splitlines()
    want = open(os.path.join(gold_dir, "demo_telolengths_all.csv")).read().splitlines()
    assert got == want
    log = open(out / "topsicle_run.log").read()
    g = json.load(open(os.path.join(gold_dir, "demo_run_log.json")))
    for key in ("patterns_line", "median_line", "asymptotic_line", "filtered_line"):
        assert g[key] in log, key
    assert "All telomere found, have a nice day." in log
    # filtered fastq: the 17 passing records, byte-identical to the input records
    filt = out / "Col-0-6909_GWHBDNP00000001.1_nano_right.fastq_trc_over_0.7.fastq"
    recs = list(seqio.read_records(str(filt)))
    src = {r.id: r for r in seqio.read_records(demo_fastq)}
    assert [r.id for r in recs] == [w.split(",")[3] for w in want[1:]]
    assert all(r.seq == src[r.id].seq and r.qual == src[r.id].qual and r.description == src[r.id].description for r in recs)
    assert os.path.exists(out / "quadfit_5mer_CCCTAAA.png")
    # second run without --override refuses (main.py:181-187)
    with pytest.raises(SystemExit):
        run_cli(engine, ["--inputDir", os.path.dirname(demo_fastq), "--outputDir", str(out), "--pattern", "CCCTAAA", "--slide", "6"])
    run_cli(engine, ["-i", os.path.dirname(demo_fastq), "-o", str(out), "--pattern", "CCCTAAA", "--slide", "6", "--override"])
    assert open(out / "telolengths_all.csv").read().splitlines() == want


def test_cli_rawcount_plot_readcheck_and_multik(engine, demo_fastq, tmp_path, demo_windows):
    meta, arrs = demo_windows
    out = tmp_path / "out2"
    rid = meta["reads"][2]["id"]
    run_cli(engine, ["-i", demo_fastq, "-o", str(out), "--pattern", "CCCTAAA", "--slide", "6", "--rawcountpattern",
                     "--plot", "--read_check", rid, "--telophrase", "5", "4"])
    rows = list(csv.reader(open(out / "telolengths_all.csv")))
    assert rows[0] == ["file_number", "phrase", "trc", "readID", "telo_length"]
    assert [r[1] for r in rows[1:]] == ["5", "4"] and all(r[3] == rid for r in rows[1:])
    assert rows[1][4] == str(meta["reads"][2]["boundary"])
    raw = list(csv.reader(open(out / "rawcount_5_1.csv")))
    assert raw[0] == ["", "tail", "position", "pattern", "count"]
    counts = np.array([int(r[4]) for r in raw[1:]]).reshape(-1, 14)
    assert np.array_equal(counts, arrs["counts_2"])
    assert os.path.exists(out / "plot_5_1.png") and os.path.exists(out / "plot_4_1.png")


def test_cli_rawcount_columnar_extension(engine, demo_fastq, tmp_path, demo_windows):
    """--rawcountformat npz: the same counts as the per-read CSVs, one file per input file and k."""
    meta, arrs = demo_windows
    out = tmp_path / "out3"
    run_cli(engine, ["-i", demo_fastq, "-o", str(out), "--pattern", "CCCTAAA", "--slide", "6", "--rawcountpattern", "--rawcountformat", "npz"])
    z = np.load(out / f"rawcount_5_{os.path.splitext(os.path.basename(demo_fastq))[0]}.npz")
    assert not [f for f in os.listdir(out) if f.startswith("rawcount_") and f.endswith(".csv")]
    ids = [str(x) for x in z["read_id"]]
    assert len(ids) == len(list(csv.reader(open(out / "telolengths_all.csv")))) - 1
    assert z["counts"].dtype == np.uint8 and z["counts"].shape[1] == 14 and int(z["slide"]) == 6
    for j, r in enumerate(meta["reads"]):
        if f"counts_{j}" in arrs and r["id"] in ids:
            i = ids.index(r["id"])
            assert np.array_equal(z["counts"][z["win_off"][i]:z["win_off"][i + 1]], arrs[f"counts_{j}"])


def test_cli_fasta_gz_input_and_default_slide(engine, tmp_path, demo_records):
    """FASTA input (wrapped lines, gz), default slide = len(pattern) (main.py:212-215)."""
    fa = tmp_path / "reads.fa.gz"
    with gzip.open(fa, "wt") as h:
        for rid, seq in demo_records[:12]:
            h.write(f">{rid} some description\n")
            for i in range(0, len(seq), 80):
                h.write(seq[i:i + 80] + "\n")
    out = tmp_path / "o"
    run_cli(engine, ["-i", str(fa), "-o", str(out), "--pattern", "AAACCCT", "--cutoff", "0.7", "0.5"])
    rows = list(csv.reader(open(out / "telolengths_all.csv")))[1:]
    assert rows and all(r[0] == "reads.fa" for r in rows)
    # filtered at min(cutoff)=0.5, reported with cutoff[0]=0.7 (main.py:56, 254-257)
    assert os.path.exists(out / "reads.fa_trc_over_0.5.fasta")
    assert "with TRC >= 0.7" in open(out / "topsicle_run.log").read()
    import topsicle_oracle as orc
    pats = orc.kmer_table("AAACCCT", 5)
    seqs = dict(demo_records)
    for r in rows:
        cs, ce = orc.trc_counts(seqs[r[3]], pats)
        call = orc.trc_call(cs, ce, pats, 7, 0.5)
        assert f"{call[2]:.3f}" == r[2]
        assert orc.step2(seqs[r[3]], call[1], pats, 100, 7, 100, 20000) == int(r[4])


def test_cli_processes_several_files_concurrently(tmp_path, gold_dir):
    """A directory of input files with an engine factory: files are processed on their own threads and
    contexts; every file contributes its rows, grouped per file, and its filtered file."""
    d = tmp_path / "many"
    d.mkdir()
    names = ["a_sample", "b_sample", "c_sample"]
    for n in names:
        shutil.copyfile(os.path.join(gold_dir, "demo_col0.fastq.gz"), d / f"{n}.fastq.gz")
    out = tmp_path / "out"
    made = []

    def factory():
        made.append(EmuEngine())
        return [made[-1]]

    args = cli.build_parser().parse_args(["-i", str(d), "-o", str(out), "--pattern", "CCCTAAA", "--slide", "6", "--threads", "3"])
    cli.tprint.logfile = cli.get_log_path(args)
    cli.analysis_run(args, engine_factory=factory)
    assert len(made) >= 2                                      # the first set + at least one worker's own
    want = open(os.path.join(gold_dir, "demo_telolengths_all.csv")).read().splitlines()[1:]
    rows = list(csv.reader(open(out / "telolengths_all.csv")))[1:]
    assert len(rows) == 3 * len(want)
    for n in names:
        mine = [",".join(r[1:]) for r in rows if r[0] == f"{n}.fastq"]
        assert mine == [w.split(",", 1)[1] for w in want], n    # same rows, same order within the file
        assert os.path.exists(out / f"{n}.fastq_trc_over_0.7.fastq")
    assert "processing 3 files, 3 at a time" in open(out / "topsicle_run.log").read()


def test_cli_several_k_in_one_pass_equals_separate_runs(engine, demo_fastq, tmp_path):
    """`--telophrase 4 5 6`: every batch is uploaded once and scanned once per k; telolengths_all.csv, the raw-count
    files and the filtered fastq are what three separate single-k runs leave behind (rows of one k before the next,
    main.py:206-235)."""
    single_rows = []
    for k in (4, 5, 6):
        out = tmp_path / f"k{k}"
        run_cli(engine, ["-i", demo_fastq, "-o", str(out), "--pattern", "CCCTAAA", "--slide", "6", "--telophrase", str(k), "--rawcountpattern"])
        single_rows += open(out / "telolengths_all.csv").read().splitlines()[1:]
    out = tmp_path / "all"
    run_cli(engine, ["-i", demo_fastq, "-o", str(out), "--pattern", "CCCTAAA", "--slide", "6", "--telophrase", "4", "5", "6", "--rawcountpattern"])
    got = open(out / "telolengths_all.csv").read().splitlines()
    assert got[0] == "file_number,phrase,trc,readID,telo_length" and got[1:] == single_rows
    for k in (4, 5, 6):
        for f in sorted(os.listdir(tmp_path / f"k{k}")):
            if f.startswith("rawcount_"):
                assert open(tmp_path / f"k{k}" / f).read() == open(out / f).read(), f
    last = [f for f in os.listdir(tmp_path / "k6") if "_trc_over_" in f][0]
    assert open(tmp_path / "k6" / last).read() == open(out / last).read()
    log = open(out / "topsicle_run.log").read()
    assert all(f"k-mer: {k}, with TRC >= 0.7" in log for k in (4, 5, 6))


def test_cli_fasta_input_several_k_keeps_the_first_k_records(engine, tmp_path, demo_records):
    """FASTA input, `--telophrase 4 6`: upstream's outer loop over k writes <name>.fasta for the first k and every later k
    finds and REUSES it (main.py:64-66), so the filtered file holds the FIRST k's passing records (for FASTQ input the file is
    rewritten per k and the last k's records stay: the test above)."""
    fa = tmp_path / "reads.fasta"
    with open(fa, "wt") as h:
        for rid, seq in demo_records:
            h.write(f">{rid}\n{seq}\n")
    outs = {}
    for tag, ks in (("k4", ["4"]), ("k6", ["6"]), ("both", ["4", "6"])):
        out = tmp_path / tag
        run_cli(engine, ["-i", str(fa), "-o", str(out), "--pattern", "CCCTAAA", "--slide", "6", "--cutoff", "0.3", "--telophrase"] + ks)
        outs[tag] = open(out / "reads_trc_over_0.3.fasta").read()
    assert outs["k4"] != outs["k6"]                     # (the two tables pass different reads at this cutoff)
    assert outs["both"] == outs["k4"]


def test_scan_jobs_tables_on_helper_contexts_equal_back_to_back(monkeypatch):
    """batch.scan_jobs with several pattern tables: job j > 0 runs on the engine's j-th helper context, which borrows the batch
    (share) and keeps its own table -- the host logic of `--telophrase 4 5 6`, here on emulated contexts: the same rows, sums
    and raw counts as the tables back to back on the one context (batch.SEQUENTIAL_TABLES), batch after batch."""
    from emu_engine import EmuEngine
    import topsicle_oracle as orc
    from topsicle_amd import batch, hiplib, synth
    eng = EmuEngine()
    prm = hiplib.make_params(no_bp=300, min_len=500, min_count=3, window=100, slide=6, trimfirst=100, maxlen=20000,
                             flags=hiplib.F_STEP1 | hiplib.F_WINDOWS | hiplib.F_BINSEG)
    jobs = [batch.Job(orc.kmer_table("CCCTAA", k), prm, want_sums=True, want_raw=(k == 5)) for k in (4, 5, 6)]
    for seed, n in ((1, 7), (2, 12)):
        bases, offsets, _ = synth.make_reads(n, 2500, "CCCTAA", seed=seed, errors=synth.ONT, tract_min=300, tract_max=1800)
        recs = type("B", (), {"bases": bases, "offsets": offsets})()
        monkeypatch.setattr(batch, "SEQUENTIAL_TABLES", True)
        seq = batch.scan_jobs(eng, recs, jobs)
        monkeypatch.setattr(batch, "SEQUENTIAL_TABLES", False)
        con = batch.scan_jobs(eng, recs, jobs)
        assert len(eng._helpers) == 2 and [h.patterns for h in eng._helpers] == [jobs[1].patterns, jobs[2].patterns]
        assert eng.patterns == jobs[0].patterns                          # (the owner keeps the first table: no switch per batch)
        for (r1, s1, w1, o1), (r2, s2, w2, o2) in zip(seq, con):
            for f in ("pass", "tail", "best_start", "best_end", "n_win", "bkp"):
                assert np.array_equal(r1[f], r2[f]), f
            assert np.array_equal(o1, o2) and np.array_equal(s1, s2)
            assert (w1 is None) == (w2 is None) and (w1 is None or np.array_equal(w1, w2))
        assert con[0][0]["pass"].sum() > 0


def test_cli_same_outputs_from_plain_gzip_and_bgzf_input(engine, tmp_path, demo_records, monkeypatch):
    """The same reads as plain FASTQ, ordinary gzip (inflated by the thread team: the option pargz_min = 0 takes that path for a small
    file too) and BGZF, with windows of inflated text far smaller than the file (many refills; the passing records of a batch are
    written from the window the batch keeps alive): telolengths_all.csv and the filtered FASTQ are byte for byte the same."""
    import struct
    import zlib
    from topsicle_amd import e2e
    rng = np.random.default_rng(2)
    fq = tmp_path / "reads.fastq"
    with open(fq, "wb") as h:
        for rid, seq in demo_records[:30]:
            q = bytes(rng.integers(35, 74, len(seq), dtype=np.uint8))
            h.write(b"@" + rid.encode() + b" extra words\n" + seq.encode() + b"\n+\n" + q + b"\n")
    d_gz, d_bg = tmp_path / "gz", tmp_path / "bg"
    d_gz.mkdir(); d_bg.mkdir()
    with open(fq, "rb") as src, gzip.open(d_gz / "reads.fastq.gz", "wb", compresslevel=1) as dst:
        dst.write(src.read())
    e2e.write_bgzf(str(d_bg / "reads.fastq.gz"), str(fq), block=20000)
    seqio.io_option("pargz_min", 0)
    seqio.io_option("bgzf_group", 150000)
    outs = {}
    for tag, path in (("plain", fq), ("gz", d_gz / "reads.fastq.gz"), ("bgzf", d_bg / "reads.fastq.gz")):
        out = tmp_path / ("o_" + tag)
        run_cli(engine, ["-i", str(path), "-o", str(out), "--pattern", "CCCTAAA", "--slide", "6", "--cutoff", "0.4"])
        rows = [r[1:] for r in csv.reader(open(out / "telolengths_all.csv"))]          # (column 0 is the file's name: reads / reads.fastq)
        filtered = [f for f in os.listdir(out) if "_trc_over_0.4" in f]
        assert len(filtered) == 1 and filtered[0].endswith(".fastq")
        outs[tag] = (rows, open(out / filtered[0], "rb").read())
    assert len(outs["plain"][0]) > 5 and len(outs["plain"][1]) > 100000
    assert outs["gz"] == outs["plain"] and outs["bgzf"] == outs["plain"]
