"""Parity of the HIP kernels (through the C ABI, on a real MI355X) with the reference-generated
goldens and the CPU oracle.  Integer work: every comparison is bit-exact."""
import csv
import os

import numpy as np
import pytest

import topsicle_oracle as orc
import oracle_c
from topsicle_amd import allsteps, hiplib, synth

pytestmark = pytest.mark.gpu
TAILV = {"forward": 0, "reverse": 1}
WIN_FLAGS = hiplib.F_WINDOWS | hiplib.F_BINSEG | hiplib.F_TAILS_IN | hiplib.F_STORE_SUMS


@pytest.fixture(scope="module")
def sc():
    s = hiplib.HipScanner(0)
    yield s
    s.close()


def test_library_reports_gfx950(sc):
    info = sc.device_info()
    assert "gfx950" in info, info


def test_synthetic_goldens(sc, synth_cases):
    meta, arrs = synth_cases
    done = 0
    for ci, c in enumerate(meta):
        pats = c["patterns"]
        if c["k"] > hiplib.MAX_K or len(pats) > hiplib.MAX_PATTERNS:
            continue
        sc.set_patterns(pats)
        bases, offsets = hiplib.pack_reads([c["seq"]])
        cs, ce = sc.trc_counts(bases, offsets, c["no_bp"])
        ws, we = orc.trc_counts(c["seq"], pats, c["no_bp"])
        assert cs[0].tolist() == ws and ce[0].tolist() == we, c["name"]
        for tail in c["tails"]:
            want = arrs[f"counts_{ci}_{tail}"].astype(np.int64)
            sums, win_off, raw = sc.window_counts(bases, offsets, [TAILV[tail]], c["W"], c["s"], c["t"], c["M"], raw=True)
            assert raw.shape[0] == want.shape[0], c["name"]
            if want.shape[0]:
                assert np.array_equal(raw, want), (c["name"], tail)
                assert np.array_equal(sums, want.sum(axis=1)), (c["name"], tail)
            sums2, _, _ = sc.window_counts(bases, offsets, [TAILV[tail]], c["W"], c["s"], c["t"], c["M"], raw=False)
            assert np.array_equal(sums2, sums), (c["name"], tail)
            bkp, gain = sc.binseg_l2(sums, win_off, len(pats))
            b = c["boundary"][tail]
            if b is None:
                assert bkp[0] == -1, c["name"]
            else:
                # (binseg_l2 hands ties to ruptures' float64 arithmetic: the reference-generated boundary, polyC's exact tie included)
                assert bkp[0] * c["s"] + c["t"] == b, c["name"]
                if c["name"] != "polyC":
                    assert bkp[0] == orc.binseg_l2_exact(want.sum(axis=1)), c["name"]
        done += 1
    assert done >= 50


def test_demo_file_end_to_end(sc, demo_records, demo_windows, gold_dir):
    """44 demo reads in one launch: the 17 golden reads pass, TRC to 3 dp, boundaries equal,
    per-window sums equal the reference's rawCountPattern sums."""
    meta, arrs = demo_windows
    pats = meta["patterns"]
    sc.set_patterns(pats)
    bases, offsets = hiplib.pack_reads([s for _, s in demo_records])
    sc.upload(0, bases, offsets)
    ratio = 1000 / 7
    min_count = max(c for c in range(1001) if not (c / ratio > 0.7))
    prm = hiplib.make_params(min_len=9000, min_count=min_count, window=100, slide=6, trimfirst=100, maxlen=20000,
                             flags=hiplib.F_STEP1 | hiplib.F_WINDOWS | hiplib.F_BINSEG | hiplib.F_STORE_SUMS)
    sc.scan(0, prm)
    sc.sync()
    res = sc.results(0)
    sums, win_off = sc.window_sums(0)
    gold = list(csv.reader(open(os.path.join(gold_dir, "demo_telolengths_all.csv"))))[1:]
    passing = [i for i in range(len(demo_records)) if res["pass"][i]]
    assert [demo_records[i][0] for i in passing] == [g[3] for g in gold]
    for j, i in enumerate(passing):
        best = res["best_start"][i] if res["tail"][i] == 0 else res["best_end"][i]
        assert f"{best / ratio:.3f}" == gold[j][2]
        assert int(res["bkp"][i]) * 6 + 100 == int(gold[j][4])
        assert np.array_equal(sums[win_off[i]:win_off[i + 1]], arrs[f"counts_{j}"].astype(np.int64).sum(axis=1))


@pytest.mark.parametrize("seed", range(8))
def test_random_vs_oracle(sc, seed):
    rng = np.random.default_rng(500 + seed)
    motif, k = [("CCCTAA", 4), ("CCCTAA", 5), ("AAACCCT", 5), ("CCCTAA", 6), ("TTAGGG", 3), ("AAACCCT", 7),
                ("CCCTAA", 4), ("TTTAGGG", 5)][seed]
    pats = orc.kmer_table(motif, k)
    sc.set_patterns(pats)
    W = int(rng.choice([100, 64, 23, 100]))
    s = int(rng.choice([6, 7, 1, 4, 16, 11]))
    t = int(rng.choice([100, 0, 17]))
    M = int(rng.choice([20000, 900, 1500]))
    seqs, tails = [], []
    for i in range(12):
        L = int(rng.integers(0, 4000))
        tract = int(rng.integers(0, max(1, L // 2)))
        ph = int(rng.integers(len(motif)))
        body = list(((motif * (tract // len(motif) + 2))[ph:ph + tract] +
                     "".join("ACGT"[x] for x in rng.integers(0, 4, max(0, L - tract))))[:L])
        for p in rng.integers(0, max(1, len(body)), len(body) // 25):
            if body:
                body[p] = "ACGTNacgtn"[int(rng.integers(10))]
        seq = "".join(body)
        if rng.random() < 0.5:
            seq = seq[::-1].translate(str.maketrans("ACGTacgt", "TGCAtgca"))
        seqs.append(seq)
        tails.append(int(rng.integers(2)))
    bases, offsets = hiplib.pack_reads(seqs)
    cs, ce = sc.trc_counts(bases, offsets)
    sums, win_off, raw = sc.window_counts(bases, offsets, tails, W, s, t, M, raw=True)
    bkp, _ = sc.binseg_l2(sums, win_off, len(pats))
    for i, seq in enumerate(seqs):
        ws, we = orc.trc_counts(seq, pats)
        assert cs[i].tolist() == ws and ce[i].tolist() == we
        _, counts = orc.window_count_matrix(seq, ["forward", "reverse"][tails[i]], pats, W, s, t, M)
        lo, hi = win_off[i], win_off[i + 1]
        assert hi - lo == counts.shape[0]
        assert np.array_equal(raw[lo:hi], counts.reshape(-1, len(pats)))
        assert np.array_equal(sums[lo:hi], counts.sum(axis=1))
        # (HipScanner.binseg_l2 = the kernel's answer, exact ties handed to ruptures' float64 arithmetic: the float64 restatement)
        want = orc.binseg_l2_numpy(counts.sum(axis=1) / len(pats))[0] if counts.shape[0] else None
        assert bkp[i] == (-1 if want is None else want)


def test_config2_batch_sample_vs_oracle(sc):
    """BASELINE config 2 shape (15 kb ONT-like reads, CCCTAA, k=4, W=100, s=6) at a size the
    oracle finishes in seconds; full fused pipeline in one launch."""
    motif, k = "CCCTAA", 4
    pats = orc.kmer_table(motif, k)
    sc.set_patterns(pats)
    bases, offsets, truth = synth.make_reads(64, 15000, motif, seed=20250920)
    sc.upload(1, bases, offsets)
    ratio = 1000 / 6
    min_count = max(c for c in range(1001) if not (c / ratio > 0.7))
    prm = hiplib.make_params(min_len=9000, min_count=min_count,
                             flags=hiplib.F_STEP1 | hiplib.F_WINDOWS | hiplib.F_BINSEG | hiplib.F_STORE_SUMS)
    sc.scan(1, prm)
    sc.sync()
    res = sc.results(1)
    sums, win_off = sc.window_sums(1)
    seqs = synth.split_reads(bases, offsets)
    for i in range(0, 64, 4):
        cs, ce = orc.trc_counts(seqs[i], pats)
        call = orc.trc_call(cs, ce, pats, 6, 0.7)
        assert bool(res["pass"][i]) == (call is not None)
        if call is None:
            continue
        assert res["tail"][i] == TAILV[call[1]] == int(truth["reverse"][i])
        _, counts = orc.window_count_matrix(seqs[i], call[1], pats, 100, 6, 100, 20000)
        assert np.array_equal(sums[win_off[i]:win_off[i + 1]], counts.sum(axis=1))
        assert res["bkp"][i] == orc.binseg_l2_exact(counts.sum(axis=1))


def test_full_size_properties(sc):
    """Size-independent properties at BASELINE config-2 scale (10k x 15 kb), no oracle:
      * reverse-complementing every read and swapping forward/reverse gives the same S_w
        when the complement k-mers are in the table (they always are, allsteps.py:115-119);
      * duplicating the batch gives identical per-read results (independence of reads);
      * boundaries land near the planted tract end for clean reads."""
    motif, k = "CCCTAA", 4
    pats = orc.kmer_table(motif, k)
    sc.set_patterns(pats)
    n = 10000
    bases, offsets, truth = synth.make_reads(n, 15000, motif, seed=20250920)
    comp = np.zeros(256, np.uint8)
    comp[list(b"ACGT")] = list(b"TGCA")
    rc = comp[bases.reshape(n, -1)[:, ::-1]].reshape(-1)
    prm = hiplib.make_params(min_len=9000, min_count=-1,
                             flags=hiplib.F_STEP1 | hiplib.F_WINDOWS | hiplib.F_BINSEG | hiplib.F_STORE_SUMS)
    sc.upload(2, bases, offsets)
    sc.scan(2, prm)
    sc.sync()
    res = sc.results(2).copy()
    sums, win_off = sc.window_sums(2)
    sc.upload(3, rc, offsets)
    sc.scan(3, prm)
    sc.sync()
    res_rc = sc.results(3)
    sums_rc, _ = sc.window_sums(3)
    nw = 2467
    assert np.array_equal(res["n_win"], res_rc["n_win"]) and res["n_win"].min() == nw == res["n_win"].max()
    strict = res["best_start"] != res["best_end"]          # ties go to 'reverse' on both strands
    assert strict.mean() > 0.99
    assert np.array_equal(res["tail"][strict], 1 - res_rc["tail"][strict])
    assert np.array_equal(res["best_start"], res_rc["best_end"]) and np.array_equal(res["best_end"], res_rc["best_start"])
    # the table holds every k-mer and its complement, so S_w is strand-symmetric window by window
    S, S_rc = sums.reshape(n, nw), sums_rc.reshape(n, nw)
    assert np.array_equal(S[strict], S_rc[strict])
    assert np.array_equal(res["bkp"][strict], res_rc["bkp"][strict])
    assert (res["tail"] == truth["reverse"]).mean() > 0.999
    # planted tract end vs called boundary (ONT-like errors)
    called = res["bkp"].astype(np.int64) * 6 + 100
    err = np.abs(called - truth["tract"])
    assert np.median(err) <= 60, np.median(err)
    # S_w range: P <= S_w <= P * floor(99 / k)
    assert sums.min() >= len(pats) and sums.max() <= len(pats) * (99 // k)


def test_cli_demo_on_gpu(tmp_path, gold_dir):
    """`topsicle` CLI end to end on the MI355X: the reference's demo CSV, log lines and filtered fastq."""
    import json
    import shutil
    from topsicle_amd import main as cli, seqio
    d = tmp_path / "in"
    d.mkdir()
    fq = d / "Col-0-6909_GWHBDNP00000001.1_nano_right.fastq.gz"
    shutil.copyfile(os.path.join(gold_dir, "demo_col0.fastq.gz"), fq)
    out = tmp_path / "out"
    cli.main(["--inputDir", str(d), "--outputDir", str(out), "--pattern", "CCCTAAA", "--slide", "6", "--rawcountpattern"])
    got = open(out / "telolengths_all.csv").read().splitlines()
    want = open(os.path.join(gold_dir, "demo_telolengths_all.csv")).read().splitlines()
    assert got == want
    log = open(out / "topsicle_run.log").read()
    g = json.load(open(os.path.join(gold_dir, "demo_run_log.json")))
    for key in ("patterns_line", "median_line", "asymptotic_line", "filtered_line"):
        assert g[key] in log, key
    recs = list(seqio.read_records(str(out / "Col-0-6909_GWHBDNP00000001.1_nano_right.fastq_trc_over_0.7.fastq")))
    assert [r.id for r in recs] == [w.split(",")[3] for w in want[1:]]
    meta = json.load(open(os.path.join(gold_dir, "demo_windows.json")))
    arrs = np.load(os.path.join(gold_dir, "demo_windows.npz"))
    raw = list(csv.reader(open(out / "rawcount_5_3.csv")))[1:]
    assert np.array_equal(np.array([int(r[4]) for r in raw]).reshape(-1, 14), arrs["counts_2"])


def test_invalid_bases_and_self_overlap_at_scale(sc):
    """Thousands of reads with N / lower case sprinkled in and a self-overlapping table (CCCTAA, k=5):
    spot-check against the oracle, and the raw-count path must agree with the sums path everywhere."""
    motif, k = "CCCTAA", 5
    pats = orc.kmer_table(motif, k)
    sc.set_patterns(pats)
    bases, offsets, truth = synth.make_reads(2000, 6000, motif, seed=99, tract_min=500, tract_max=3000)
    rng = np.random.default_rng(1)
    b = bases.copy()
    pos = rng.integers(0, b.size, b.size // 200)
    b[pos] = np.frombuffer(b"NnacgtRY", dtype=np.uint8)[rng.integers(0, 8, pos.size)]
    sc.upload(4, b, offsets)
    prm = hiplib.make_params(min_len=1000, min_count=-1, flags=hiplib.F_STEP1 | hiplib.F_WINDOWS | hiplib.F_BINSEG | hiplib.F_STORE_SUMS)
    sc.scan(4, prm)
    sc.sync()
    res = sc.results(4).copy()
    sums, win_off = sc.window_sums(4)
    prm.flags |= hiplib.F_STORE_RAW
    sc.scan(4, prm)
    sc.sync()
    raw, _ = sc.window_raw(4)
    sums2, _ = sc.window_sums(4)
    assert np.array_equal(sums, sums2) and np.array_equal(raw.astype(np.int64).sum(axis=1), sums)
    seqs = synth.split_reads(b, offsets)
    for i in range(0, 2000, 97):
        tail = ["forward", "reverse"][res["tail"][i]]
        cs, ce = orc.trc_counts(seqs[i], pats)
        assert (res["best_start"][i], res["best_end"][i]) == (max(cs), max(ce))
        _, counts = orc.window_count_matrix(seqs[i], tail, pats, 100, 6, 100, 20000)
        assert np.array_equal(raw[win_off[i]:win_off[i + 1]], counts)
        assert res["bkp"][i] == orc.binseg_l2_exact(counts.sum(axis=1))


@pytest.mark.parametrize("k,with_n", [(5, True), (4, True), (6, False)])
def test_run_to_run_determinism(sc, k, with_n):
    """Same batch scanned repeatedly must give bit-identical window sums (guards against scheduling /
    uninitialised-LDS hazards that only show on the device)."""
    motif = "CCCTAA"
    sc.set_patterns(orc.kmer_table(motif, k))
    bases, offsets, _ = synth.make_reads(1500, 6000, motif, seed=7, tract_min=500, tract_max=3000)
    b = bases.copy()
    if with_n:
        rng = np.random.default_rng(2)
        pos = rng.integers(0, b.size, b.size // 200)
        b[pos] = np.frombuffer(b"NnacgtRY", dtype=np.uint8)[rng.integers(0, 8, pos.size)]
    sc.upload(5, b, offsets)
    prm = hiplib.make_params(min_len=1000, min_count=-1, flags=hiplib.F_STEP1 | hiplib.F_WINDOWS | hiplib.F_BINSEG | hiplib.F_STORE_SUMS)
    sc.scan(5, prm)
    sc.sync()
    ref, _ = sc.window_sums(5)
    ref_res = sc.results(5).copy()
    for _ in range(4):
        sc.scan(5, prm)
        sc.sync()
        s, _ = sc.window_sums(5)
        assert np.array_equal(s, ref)
        assert np.array_equal(sc.results(5)["bkp"], ref_res["bkp"])


@pytest.mark.gpu
@pytest.mark.parametrize("motif,k,slide", [("CCCTAACCTA", 8, 10), ("TTAGGGTTAGGCA", 11, 6), ("AAAACCCCTT", 9, 7)])
def test_long_kmers_hashed_table(sc, motif, k, slide):
    """k > 7 goes through the perfect-hash table of the generic kernel: counts bit-exact against the oracle."""
    rng = np.random.default_rng(k)
    pats = orc.kmer_table(motif, k)
    sc.set_patterns(pats)
    seqs = []
    for i in range(40):
        L = int(rng.integers(1200, 6000))
        tract = int(rng.integers(300, 1100))
        body = list((motif * (tract // len(motif) + 2))[:tract] + "".join("ACGT"[x] for x in rng.integers(0, 4, L - tract)))
        for p in rng.integers(0, L, L // 30):
            body[p] = "ACGTNacgt"[int(rng.integers(9))]
        s = "".join(body)
        seqs.append(s if i & 1 else s[::-1])
    bases, offsets = hiplib.pack_reads(seqs)
    sc.upload(5, bases, offsets)
    prm = hiplib.make_params(min_len=0, min_count=-1, window=100, slide=slide,
                             flags=hiplib.F_STEP1 | hiplib.F_WINDOWS | hiplib.F_BINSEG | hiplib.F_STORE_SUMS | hiplib.F_STORE_RAW)
    sc.scan(5, prm)
    sc.sync()
    res = sc.results(5).copy()
    raw, win_off = sc.window_raw(5)
    sums, _ = sc.window_sums(5)
    cs_all, ce_all = sc.batch_trc_counts(5)
    for i, seq in enumerate(seqs):
        cs, ce = orc.trc_counts(seq, pats)
        assert cs_all[i].tolist() == cs and ce_all[i].tolist() == ce
        tail = ["forward", "reverse"][res["tail"][i]]
        _, counts = orc.window_count_matrix(seq, tail, pats, 100, slide, 100, 20000)
        assert np.array_equal(raw[win_off[i]:win_off[i + 1]], counts.reshape(-1, len(pats)))
        assert np.array_equal(sums[win_off[i]:win_off[i + 1]], counts.sum(axis=1))
        want = orc.binseg_l2_exact(counts.sum(axis=1)) if counts.shape[0] >= 7 else None
        assert res["bkp"][i] == (-1 if want is None else want)


@pytest.mark.gpu
def test_randomised_parameter_sweep():
    """tests/gpu_fuzz.py: random motifs / k / window / slide / trim / maxlen / jump / min_size / no_bp and reads
    with errors, N, lower case and both strands, checked against the C oracle (it found the jump = 1 bug)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "gpu_fuzz.py"), "150", "11"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_cli_several_files_concurrently_on_gpu(tmp_path, gold_dir):
    """Several input files: each is processed on its own host thread with its own context on the same GPU."""
    import shutil
    from topsicle_amd import main as cli
    d = tmp_path / "many"
    d.mkdir()
    names = ["a_sample", "b_sample", "c_sample", "d_sample"]
    for n in names:
        shutil.copyfile(os.path.join(gold_dir, "demo_col0.fastq.gz"), d / f"{n}.fastq.gz")
    out = tmp_path / "out"
    cli.main(["--inputDir", str(d), "--outputDir", str(out), "--pattern", "CCCTAAA", "--slide", "6", "--threads", "4"])
    want = open(os.path.join(gold_dir, "demo_telolengths_all.csv")).read().splitlines()[1:]
    rows = list(csv.reader(open(out / "telolengths_all.csv")))[1:]
    assert len(rows) == len(names) * len(want)
    for n in names:
        assert [",".join(r[1:]) for r in rows if r[0] == f"{n}.fastq"] == [w.split(",", 1)[1] for w in want], n
    assert "processing 4 files, 4 at a time" in open(out / "topsicle_run.log").read()


@pytest.mark.gpu
def test_error_paths_are_loud(sc):
    """Misuse of the C ABI returns an error code + message (the Python layer raises); nothing is silently skipped."""
    with pytest.raises(hiplib.TopsicleHipError):
        sc.set_patterns(["ACGTACGTACGTACGT"])                 # k = 16 > TPS_MAX_K
    with pytest.raises(hiplib.TopsicleHipError):
        sc.set_patterns(["ACGN"])                             # non-ACGT letter in a pattern
    with pytest.raises(hiplib.TopsicleHipError):
        sc.set_patterns([f"{'ACGT'[i % 4]}{'ACGT'[(i // 4) % 4]}{'ACGT'[(i // 16) % 4]}A" for i in range(40)])   # > 31 patterns
    sc.set_patterns(orc.kmer_table("CCCTAA", 4))
    with pytest.raises(hiplib.TopsicleHipError):
        sc.scan(9, hiplib.make_params())                      # slot never uploaded
    bases, offsets = hiplib.pack_reads(["ACGT" * 500])
    sc.upload(9, bases, offsets)
    for bad in (dict(window=0), dict(slide=0), dict(jump=0), dict(slide=5000), dict(window=70000)):
        with pytest.raises(hiplib.TopsicleHipError):
            sc.scan(9, hiplib.make_params(**bad))
    with pytest.raises(hiplib.TopsicleHipError):
        sc.scan(9, hiplib.make_params(flags=hiplib.F_WINDOWS | hiplib.F_TAILS_IN))   # tails promised but never set
    with pytest.raises(hiplib.TopsicleHipError):
        hiplib.HipScanner(99)                                 # no such device
    sc.scan(9, hiplib.make_params(min_len=0, min_count=-1))   # and the context is still usable afterwards
    sc.sync()
    assert sc.results(9)["n_win"][0] == hiplib.window_count(2000, 100, 6, 100, 20000)


@pytest.mark.gpu
@pytest.mark.parametrize("motif,k,slide,units", [
    ("CCCTAA", 5, 6, ["CTAA", "GATT"]),
    ("CCCTAA", 6, 6, ["CCTAA", "GGATT", "CTAAC"]),
    ("CCCTAA", 6, 5, ["CCTAA"]),
    ("CCCTAA", 6, 8, ["CCTAA"]),
    ("TTTAGGG", 7, 6, ["TTTAGG", "AAATCC"]),
    ("TTAGGG", 5, 5, ["TTAG", "AATC"]),
    ("CCCTAA", 4, 6, []),
    ("AAACCCT", 5, 7, []),
])
def test_per_pattern_tiles_chains_and_raw_rows(sc, motif, k, slide, units):
    """The per-pattern tiles (exact counts without recounting: canonical picks, start skips, chain-parity repairs,
    look-back walks; raw rows staged through LDS) on reads full of deletions and runs of the k-mers' own period."""
    from test_emulation import _pp_reads
    rng = np.random.default_rng(sum(map(ord, motif)) * 1000 + 10 * k + slide + 7)
    pats, seqs = _pp_reads(rng, motif, k, 24, 14000, units)
    seqs[5] = seqs[5][:4000] + "N" + seqs[5][4001:]
    tails = [int(x) for x in rng.integers(0, 2, len(seqs))]
    sc.set_patterns(pats)
    bases, offsets = hiplib.pack_reads(seqs)
    sums, win_off, raw = sc.window_counts(bases, offsets, tails, 100, slide, 100, 20000, raw=True)
    sums2, win_off2, _ = sc.window_counts(bases, offsets, tails, 100, slide, 100, 20000, raw=False)
    bkp, _ = sc.binseg_l2(sums, win_off, len(pats))
    assert np.array_equal(win_off, win_off2) and np.array_equal(sums, sums2)
    for i, seq in enumerate(seqs):
        _, counts = orc.window_count_matrix(seq, ["forward", "reverse"][tails[i]], pats, 100, slide, 100, 20000)
        lo, hi = win_off[i], win_off[i + 1]
        assert hi - lo == counts.shape[0]
        assert np.array_equal(raw[lo:hi], counts.reshape(-1, len(pats))), i
        assert np.array_equal(sums[lo:hi], counts.sum(axis=1)), i
        assert bkp[i] == orc.binseg_l2_numpy(counts.sum(axis=1) / len(pats))[0]


@pytest.mark.gpu
def test_per_pattern_tiles_adversarial_sweep():
    """tests/gpu_fuzz_pp.py: deletion-ridden repeats and runs of the k-mers' own period over 15 tables / slides, raw rows
    and sums, both tails, against the C oracle."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "gpu_fuzz_pp.py"), "90", "3"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_raw_rows_for_any_number_of_patterns():
    """tests/gpu_raw_any_p.py: hand-made pattern lists of 1 .. 14 patterns (odd row lengths leave the per-pattern tiles byte by
    byte, others in 16-bit units or packed), raw rows and sums against the oracle."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "gpu_raw_any_p.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("k", [4, 5, 6])
def test_full_size_raw_count_properties(sc, k):
    """BASELINE config-5 shape (25 kb ONT reads, --telophrase 4 5 6 --rawcountpattern), 4000 reads per k, no oracle at this
    size: the raw rows sum to S_w window by window; every count is >= 1 (`matches or 1`) and <= the non-overlapping
    maximum; the strand-swapped batch gives the rows with pattern p and its complement exchanged; a few reads are checked
    against the C oracle."""
    motif = "CCCTAA"
    pats = orc.kmer_table(motif, k)
    P = len(pats)
    sc.set_patterns(pats)
    n, L = 4000, 25000
    bases, offsets, _ = synth.make_reads(n, L, motif, seed=20250919 + 4)
    prm = hiplib.make_params(min_len=9000, min_count=-1,
                             flags=hiplib.F_STEP1 | hiplib.F_WINDOWS | hiplib.F_BINSEG | hiplib.F_STORE_SUMS | hiplib.F_STORE_RAW)
    sc.upload(4, bases, offsets)
    sc.scan(4, prm)
    sc.sync()
    res = sc.results(4).copy()
    sums, win_off = sc.window_sums(4)
    raw, _ = sc.window_raw(4)
    raw = raw.reshape(-1, P)
    assert raw.shape[0] == sums.shape[0] == win_off[-1] == n * 3301
    assert np.array_equal(raw.sum(axis=1, dtype=np.int64), sums)
    assert raw.min() >= 1 and raw.max() <= (99 - k) // k + 1
    # strand symmetry: complement k-mers are the second half of the table (allsteps.py:115-119)
    comp = np.zeros(256, np.uint8)
    comp[list(b"ACGT")] = list(b"TGCA")
    rc = comp[bases.reshape(n, -1)[:, ::-1]].reshape(-1)
    sc.upload(5, rc, offsets)
    sc.scan(5, prm)
    sc.sync()
    res_rc = sc.results(5)
    raw_rc, _ = sc.window_raw(5)
    raw_rc = raw_rc.reshape(-1, P)
    strict = np.repeat(res["best_start"] != res["best_end"], 3301)
    half = P // 2
    swapped = np.concatenate([raw_rc[:, half:], raw_rc[:, :half]], axis=1)
    assert np.array_equal(raw[strict], swapped[strict])
    assert np.array_equal(res["tail"][res["best_start"] != res["best_end"]], 1 - res_rc["tail"][res["best_start"] != res["best_end"]])
    for i in (0, 1, 1999, 3999):
        seq = bytes(bases[offsets[i]:offsets[i + 1]]).decode()
        _, counts = orc.window_count_matrix(seq, ["forward", "reverse"][int(res["tail"][i])], pats, 100, 6, 100, 20000)
        assert np.array_equal(raw[win_off[i]:win_off[i + 1]], counts)


@pytest.mark.gpu
@pytest.mark.parametrize("motif,slide,W", [("AAACCCT", 7, 100), ("CCCTAAA", 6, 100), ("AAACCCT", 5, 100), ("CCCTAAA", 8, 260)])
def test_k5_pair_table_kernels(sc, motif, slide, W):
    """k = 5 tables without self-overlap, sums only: the planner picks the 16-bit pair table and 8-wave workgroups (kernels _s*q);
    step 1, window sums and the boundary bit-exact against the oracle -- reads of several tiles, both strands, N and lower case."""
    k = 5
    pats = orc.kmer_table(motif, k)
    sc.set_patterns(pats)
    rng = np.random.default_rng(slide * 17 + W)
    comp = str.maketrans("ACGTacgt", "TGCAtgca")
    seqs = []
    for i in range(48):
        L = int(rng.integers(3000, 12000))
        tract = int(rng.integers(600, 2900))
        body = list((motif * (tract // len(motif) + 2))[:tract] + "".join("ACGT"[x] for x in rng.integers(0, 4, L - tract)))
        for p in rng.integers(0, L, L // 14):
            body[p] = "ACGT"[int(rng.integers(4))]
        if i % 5 == 3:
            body[int(rng.integers(1200, L))] = "N"
        if i % 7 == 4:
            body[50:400] = [c.lower() for c in body[50:400]]
        sq = "".join(body)
        seqs.append(sq if i % 2 == 0 else sq[::-1].translate(comp))
    bases, offsets = hiplib.pack_reads(seqs)
    sc.upload(6, bases, offsets)
    prm = hiplib.make_params(min_len=0, min_count=-1, window=W, slide=slide,
                             flags=hiplib.F_STEP1 | hiplib.F_WINDOWS | hiplib.F_BINSEG | hiplib.F_STORE_SUMS)
    sc.scan(6, prm)
    sc.sync()
    info = sc.kernel_info(6)
    assert info.startswith("tps_scan_kernel_s%dq " % slide) and "waves_per_wg=8" in info, info
    res = sc.results(6).copy()
    sums, win_off = sc.window_sums(6)
    cs_all, ce_all = sc.batch_trc_counts(6)
    for i, seq in enumerate(seqs):
        cs, ce = orc.trc_counts(seq, pats)
        assert cs_all[i].tolist() == cs and ce_all[i].tolist() == ce, i
        tail = ["forward", "reverse"][res["tail"][i]]
        _, counts = orc.window_count_matrix(seq, tail, pats, W, slide, 100, 20000)
        assert np.array_equal(sums[win_off[i]:win_off[i + 1]], counts.sum(axis=1)), i
        want = orc.binseg_l2_exact(counts.sum(axis=1)) if counts.shape[0] >= 7 else None
        assert res["bkp"][i] == (-1 if want is None else want), i


@pytest.mark.gpu
# (round 5: _s6r carries the 4 KB pair table of fields -- 5 waves per SIMD by registers either way; _s6sol / _s6so keep their lane totals in the pad
# words (no XF array); _s6so: 80 VGPRs + the slimmer slice = three 8-wave workgroups, 6 waves per SIMD)
@pytest.mark.parametrize("k,flags,kernel", [(4, 0, "tps_scan_kernel_s6p lds=23936 wgs_per_cu=6 waves_per_wg=4"),
                                           (4, hiplib.F_STORE_RAW, "tps_scan_kernel_s6r lds=27264 wgs_per_cu=5 waves_per_wg=4"),
                                           (5, 0, "tps_scan_kernel_s6sol lds=24192 wgs_per_cu=6 waves_per_wg=4"),
                                           (5, hiplib.F_STORE_RAW, "tps_scan_kernel_s6sor lds=26240 wgs_per_cu=6 waves_per_wg=4"),
                                           (6, 0, "tps_scan_kernel_s6so lds=52480 wgs_per_cu=3 waves_per_wg=8"),
                                           (6, hiplib.F_STORE_RAW, "tps_scan_kernel_s6sorh lds=30336 wgs_per_cu=5 waves_per_wg=4")])
def test_planned_launch_shapes_of_the_benchmark_tables(sc, k, flags, kernel):
    """The kernel, LDS bytes and workgroup shape the planner picks for CCCTAA at k = 4, 5, 6 on a clean batch (BASELINE configs[1] and
    [4]): what the profiles and DESIGN quote.  (A planner edit for the k = 5 pair tables once moved the 8 KB tables of the k = 6
    kernels to 8-wave workgroups unnoticed: 493 -> 530 us per three-k step.)"""
    sc.set_patterns(orc.kmer_table("CCCTAA", k))
    bases, offsets, _ = synth.make_reads(64, 12000, "CCCTAA", seed=3)
    sc.upload(7, bases, offsets)
    prm = hiplib.make_params(min_len=0, min_count=-1, window=100, slide=6,
                             flags=hiplib.F_STEP1 | hiplib.F_WINDOWS | hiplib.F_BINSEG | hiplib.F_STORE_SUMS | flags)
    sc.scan(7, prm)
    sc.sync()
    assert sc.kernel_info(7) == kernel


@pytest.mark.gpu
@pytest.mark.parametrize("slide,k,motif", [(3, 4, "CCCTAA"), (4, 4, "CCCTAA"), (9, 4, "CCCTAA"), (10, 4, "CCCTAA"), (11, 4, "TTAGGG"), (12, 4, "CCCTAA"), (10, 5, "AAACCCT"), (12, 3, "TTAGG")])
def test_default_kernels_other_slides(sc, slide, k, motif):
    """Slides 4 and 9 .. 12 on the default kernels (sums only, no self-overlapping k-mer; `_s<slide>` / `_s<slide>p`): step 1, window
    sums and the boundary bit-exact against the oracle -- several tiles, both strands, N and lower case."""
    pats = orc.kmer_table(motif, k)
    sc.set_patterns(pats)
    rng = np.random.default_rng(slide * 19 + k)
    comp = str.maketrans("ACGTacgt", "TGCAtgca")
    seqs = []
    for i in range(40):
        L = int(rng.integers(3000, 16000))
        tract = int(rng.integers(600, 2900))
        body = list((motif * (tract // len(motif) + 2))[:tract] + "".join("ACGT"[x] for x in rng.integers(0, 4, L - tract)))
        for p in rng.integers(0, L, L // 14):
            body[p] = "ACGT"[int(rng.integers(4))]
        if i % 5 == 3:
            body[int(rng.integers(1200, L))] = "N"
        if i % 7 == 4:
            body[50:400] = [c.lower() for c in body[50:400]]
        sq = "".join(body)
        seqs.append(sq if i % 2 == 0 else sq[::-1].translate(comp))
    bases, offsets = hiplib.pack_reads(seqs)
    sc.upload(6, bases, offsets)
    prm = hiplib.make_params(min_len=0, min_count=-1, window=100, slide=slide,
                             flags=hiplib.F_STEP1 | hiplib.F_WINDOWS | hiplib.F_BINSEG | hiplib.F_STORE_SUMS)
    sc.scan(6, prm)
    sc.sync()
    info = sc.kernel_info(6)
    assert info.startswith("tps_scan_kernel_s%d" % slide) and not info.startswith("tps_scan_kernel "), info
    res = sc.results(6).copy()
    sums, win_off = sc.window_sums(6)
    cs_all, ce_all = sc.batch_trc_counts(6)
    for i, seq in enumerate(seqs):
        cs, ce = orc.trc_counts(seq, pats)
        assert cs_all[i].tolist() == cs and ce_all[i].tolist() == ce, i
        tail = ["forward", "reverse"][res["tail"][i]]
        _, counts = orc.window_count_matrix(seq, tail, pats, 100, slide, 100, 20000)
        assert np.array_equal(sums[win_off[i]:win_off[i + 1]], counts.sum(axis=1)), i
        want = orc.binseg_l2_exact(counts.sum(axis=1)) if counts.shape[0] >= 7 else None
        assert res["bkp"][i] == (-1 if want is None else want), i


@pytest.mark.gpu
def test_sixteen_patterns_take_the_fused_kernels_on_gpu(sc):
    """Round 5: an 8-letter motif at the reference's default k = len - 2 has sixteen patterns -- TTTTAGGG at k = 6,
    default slide 8, where TAGGGT / ATCCCA overlap themselves (period 5).  A clean batch takes the chain-corrected fused tiles
    (tps_scan_kernel_s8so; the generic kernel before: 398 -> 283 us per 10 000 x 15 kb reads), a batch with a non-ACGT letter keeps the
    generic kernel (the fallback tile's flag bit is the sixteenth pattern); a table of sixteen without self-overlap (CCCTAAAA) takes
    the default kernels.  Every window sum (per-read checksums), tails and change points against oracle.c."""
    for motif, k, slide, clean_kernel in (("TTTTAGGG", 6, 8, "tps_scan_kernel_s8so"), ("CCCTAAAA", 6, 8, "tps_scan_kernel_s8so"), ("AACCGGTT", 6, 8, "tps_scan_kernel_s8")):
        pats = orc.kmer_table(motif, k)
        assert len(pats) == 16
        sc.set_patterns(pats)
        bases, offsets, _ = synth.make_reads(500, 12000, motif, seed=16 + k, errors=synth.ONT)
        prm = hiplib.make_params(min_len=9000, min_count=allsteps.min_count_for_cutoff(0.7, 1000 / len(motif), 1000), slide=slide,
                                 flags=hiplib.F_STEP1 | hiplib.F_WINDOWS | hiplib.F_BINSEG | hiplib.F_STORE_SUMS)
        for dirty in (False, True):
            b = bases.copy()
            if dirty:
                b[offsets[7] + 3000] = ord("N")
            sc.upload(5, b, offsets)
            sc.scan(5, prm)
            sc.sync()
            name = sc.kernel_info(5).split()[0]
            assert name == (clean_kernel if (not dirty or clean_kernel == "tps_scan_kernel_s8") else "tps_scan_kernel"), (motif, dirty, name)
            res = sc.results(5)
            sums, win_off = sc.window_sums(5)
            out, ck = oracle_c.batch_ck(b, offsets, pats, len(motif), 1000, 9000, 0.7, 100, slide, 100, 20000, threads=8)
            got = oracle_c.checksums(sums, win_off)
            assert int(res["pass"].sum()) > 300
            for i in range(len(res)):
                assert bool(out[i, 0]) == bool(res["pass"][i]), (motif, dirty, i)
                if out[i, 0]:
                    assert int(out[i, 1]) == int(res["tail"][i]) and int(out[i, 5]) == int(res["bkp"][i]) and int(ck[i, 0]) == int(got[i]), (motif, dirty, i)


@pytest.mark.gpu
@pytest.mark.parametrize("k,raw", [(4, False), (4, True), (6, False), (5, True)])
def test_dispatch_order_changes_no_result(sc, k, raw):
    """Round 5: wave slots take the reads of a ragged batch longest first (tps::plan_dispatch_order; ScanArgs::order).  Every output
    array of such a scan equals the scan in file order (debug option "file_order"), and a sample of reads the oracle."""
    motif = "CCCTAA"
    pats = orc.kmer_table(motif, k)
    bases, offsets, _ = synth.make_ragged_reads(3000, motif, 505 + k, n_frac=0.0002)
    prm = hiplib.make_params(min_len=9000, min_count=allsteps.min_count_for_cutoff(0.7, 1000 / len(motif), 1000), slide=6,
                             flags=hiplib.F_STEP1 | hiplib.F_WINDOWS | hiplib.F_BINSEG | hiplib.F_STORE_SUMS | (hiplib.F_STORE_RAW if raw else 0))
    got = {}
    for file_order in (1, 0):
        s = hiplib.HipScanner(0)
        try:
            s.debug_option("file_order", file_order)
            s.set_patterns(pats)
            s.upload(0, bases, offsets)
            s.scan(0, prm)
            s.sync()
            got[file_order] = dict(res=s.results(0).copy(), sums=s.window_sums(0), trc=s.batch_trc_counts(0), raw=s.window_raw(0) if raw else None)
        finally:
            s.close()
    a, b = got[1], got[0]
    assert a["res"].tobytes() == b["res"].tobytes()
    assert np.array_equal(a["sums"][1], b["sums"][1])
    assert np.array_equal(a["trc"][0], b["trc"][0]) and np.array_equal(a["trc"][1], b["trc"][1])
    res, (sums, win_off) = b["res"], b["sums"]
    # (the window regions of reads that do not pass are never written: only the passing reads' are compared)
    keep = np.zeros(int(win_off[-1]), bool)
    for i in np.nonzero(res["pass"])[0]:
        keep[win_off[i]:win_off[i + 1]] = True
    assert np.array_equal(a["sums"][0][keep], b["sums"][0][keep])
    if raw:
        P = len(pats)
        assert np.array_equal(a["raw"][0].reshape(-1, P)[keep], b["raw"][0].reshape(-1, P)[keep])
    assert 500 < int(res["pass"].sum()) < 2900
    out, ck = oracle_c.batch_ck(bases, offsets, pats, len(motif), 1000, 9000, 0.7, 100, 6, 100, 20000, threads=8)
    chk = oracle_c.checksums(sums, win_off)
    for i in range(len(res)):
        assert bool(out[i, 0]) == bool(res["pass"][i]), i
        if out[i, 0]:
            assert int(out[i, 1]) == int(res["tail"][i]) and int(out[i, 5]) == int(res["bkp"][i]) and int(ck[i, 0]) == int(chk[i]), i


@pytest.mark.gpu
@pytest.mark.parametrize("motif,k,slide,raw,kernel", [
    ("CCCTAA", 4, 10, True, "tps_scan_kernel_s5r every 2nd window"), ("CCCTAA", 4, 12, True, "tps_scan_kernel_s6r every 2nd window"),
    ("CCCTAA", 4, 15, True, "tps_scan_kernel_s5r every 3rd window"), ("CCCTAA", 6, 10, False, "tps_scan_kernel_s5so every 2nd window"),
    ("CCCTAA", 6, 12, True, "tps_scan_kernel_s6sorh every 2nd window"), ("CCCTAA", 5, 14, True, "tps_scan_kernel_s7sor every 2nd window"),
    ("CCCTAA", 4, 20, False, "tps_scan_kernel_s10p every 2nd window"), ("AAACCCT", 5, 28, False, "tps_scan_kernel_s7q every 4th window"),
    ("TTTTAGGG", 6, 16, False, "tps_scan_kernel_s8so every 2nd window"),
    ("CCCTAA", 4, 10, "clean", "tps_scan_kernel_s5r every 2nd window"), ("CCCTAA", 6, 12, "clean", "tps_scan_kernel_s6sorh every 2nd window"),
    ("CCCTAA", 5, 10, "clean", "tps_scan_kernel_s5sor every 2nd window")])
def test_strided_scans_keep_every_mth_window_of_a_fused_kernel(sc, motif, k, slide, raw, kernel):
    """Round 5: `--slide 10 --rawcountpattern`, `--slide 10 --telophrase 6` and the like have no fused kernel of their
    own; their windows are every m-th window of a slide that has (tps::stride_base): the launch shape is asserted, every window sum
    (per-read checksums), tails, change points against oracle.c, raw rows against the Python oracle on a sample, and the whole scan
    against the generic kernel (debug option "no_stride") byte for byte -- ragged reads with a few N."""
    pats = orc.kmer_table(motif, k)
    P = len(pats)
    # (sixteen patterns with a self-overlap take the fused tiles on clean batches only: no N for that table)
    # raw == "clean": no N in the batch -- at twice the base slide the tiles then store every second raw row themselves (ScanArgs::raw_m)
    bases, offsets, _ = synth.make_ragged_reads(600, motif, 700 + slide + k, n_frac=0.0 if (P == 16 or raw == "clean") else 0.0002, len_mu=9.5, len_sigma=0.5, max_len=26000)
    prm = hiplib.make_params(min_len=9000, min_count=allsteps.min_count_for_cutoff(0.7, 1000 / len(motif), 1000), slide=slide,
                             flags=hiplib.F_STEP1 | hiplib.F_WINDOWS | hiplib.F_BINSEG | hiplib.F_STORE_SUMS | (hiplib.F_STORE_RAW if raw else 0))
    got = {}
    for no_stride in (0, 1):
        s = hiplib.HipScanner(0)
        try:
            s.debug_option("no_stride", no_stride)
            s.set_patterns(pats)
            s.upload(0, bases, offsets)
            for _ in range(2):                                        # (the second scan reuses the cached plan and the borrowed batch)
                s.scan(0, prm)
            s.sync()
            name = s.kernel_info(0).split(" lds=")[0]
            assert name == (kernel if not no_stride else "tps_scan_kernel"), name
            got[no_stride] = dict(res=s.results(0).copy(), sums=s.window_sums(0), trc=s.batch_trc_counts(0), raw=s.window_raw(0) if raw else None)
        finally:
            s.close()
    a, b = got[0], got[1]
    for f in ("pass", "tail", "n_win", "bkp", "best_start", "best_end", "best_start_idx", "best_end_idx", "flags"):
        assert np.array_equal(a["res"][f], b["res"][f]), f
    assert np.array_equal(a["sums"][1], b["sums"][1])
    assert np.array_equal(a["trc"][0], b["trc"][0]) and np.array_equal(a["trc"][1], b["trc"][1])
    res, (sums, win_off) = a["res"], a["sums"]
    # (the window regions of reads that do not pass are never written by either route: only the passing reads' are compared)
    keep = np.zeros(int(win_off[-1]), bool)
    for i in np.nonzero(res["pass"])[0]:
        keep[win_off[i]:win_off[i + 1]] = True
    assert np.array_equal(a["sums"][0][keep], b["sums"][0][keep])
    if raw:
        assert np.array_equal(a["raw"][0].reshape(-1, P)[keep], b["raw"][0].reshape(-1, P)[keep])
        rows = a["raw"][0].reshape(-1, P)
        seqs = synth.split_reads(bases, offsets)
        done = 0
        for i in np.nonzero(res["pass"])[0][:5]:
            _, want = orc.window_count_matrix(seqs[i], ["forward", "reverse"][int(res["tail"][i])], pats, 100, slide, 100, 20000)
            assert np.array_equal(rows[win_off[i]:win_off[i + 1]], want), i
            done += 1
        assert done >= 3
    assert 200 < int(res["pass"].sum())
    out, ck = oracle_c.batch_ck(bases, offsets, pats, len(motif), 1000, 9000, 0.7, 100, slide, 100, 20000, threads=8)
    chk = oracle_c.checksums(sums, win_off)
    for i in range(len(res)):
        assert bool(out[i, 0]) == bool(res["pass"][i]), i
        if out[i, 0]:
            assert int(out[i, 1]) == int(res["tail"][i]) and int(out[i, 5]) == int(res["bkp"][i]) and int(ck[i, 0]) == int(chk[i]), i
