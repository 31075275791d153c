"""GPU parity on what real input looks like -- RAGGED batches at scale -- and the multi-context pipeline that no hardware run
had exercised with more than two contexts:

  * >= 5 000 reads of log-normal length (60 b .. 60 kb: some shorter than no_bp, some shorter than trimfirst + window, many
    longer than maxlengthtelo), N and lower case sprinkled in, both strands, through the host packer and the packed upload:
    every read against the float64 pipeline of the C oracle (pass, tail, best count, n_win, change-point), every 50th read's
    window sums -- and raw rows, with and without self-overlapping k-mers -- window by window;
  * batch.EnginePool with four contexts on this box's GPU ("2 GPUs x 2 contexts") over a ragged multi-batch file: results in
    input order and equal to a one-context run; the pinned staging pool is allocated once and reused file after file.

Everything goes through the C ABI on a real MI355X.  Integer work is compared bit-exact.
"""
import os

import numpy as np
import pytest

import oracle_c
import topsicle_oracle as orc
from topsicle_amd import allsteps, batch, hiplib, seqio, synth

pytestmark = pytest.mark.gpu
TAILS = ["forward", "reverse"]
FULL = hiplib.F_STEP1 | hiplib.F_WINDOWS | hiplib.F_BINSEG | hiplib.F_STORE_SUMS


@pytest.fixture(scope="module")
def sc():
    s = hiplib.HipScanner(0)
    yield s
    s.close()


def _ragged(n, motif, seed, errors=synth.ONT):
    """n reads of log-normal length plus a sprinkle of very short ones (below no_bp, below trimfirst + window)."""
    b1, o1, t1 = synth.make_ragged_reads(n - 60, motif, seed, errors=errors, n_frac=0.0003, lower_frac=0.1)
    b2, o2, t2 = synth.make_ragged_reads(60, motif, seed + 1, errors=errors, len_mu=5.6, len_sigma=0.7, min_len=60, max_len=2000,
                                         tract_min=50, tract_max=1500, n_frac=0.001, lower_frac=0.2)
    # the short ones go in between, not at the end: tiles of long and short reads next to each other in one launch
    rng = np.random.default_rng(seed)
    order = rng.permutation(n)
    lens = np.concatenate([np.diff(o1), np.diff(o2)])
    starts = np.concatenate([o1[:-1], o2[:-1] + o1[-1]])
    allb = np.concatenate([b1, b2])
    offsets = np.zeros(n + 1, np.int64)
    np.cumsum(lens[order], out=offsets[1:])
    bases = np.empty(int(offsets[-1]), np.uint8)
    for j, i in enumerate(order):
        bases[offsets[j]:offsets[j + 1]] = allb[starts[i]:starts[i] + lens[i]]
    return bases, offsets


def _params(motif, slide, flags=FULL, cutoff=0.5, min_len=1000):
    return hiplib.make_params(no_bp=1000, min_len=min_len, min_count=allsteps.min_count_for_cutoff(cutoff, 1000 / len(motif), 1000),
                              window=100, slide=slide, trimfirst=100, maxlen=20000, flags=flags)


@pytest.mark.parametrize("motif,k,slide,raw", [("CCCTAA", 4, 6, False), ("CCCTAA", 4, 6, True), ("CCCTAA", 6, 6, True), ("AAACCCT", 5, 7, False)])
def test_ragged_batch_at_scale_vs_oracle(sc, motif, k, slide, raw):
    n = 5000
    pats = orc.kmer_table(motif, k)
    sc.set_patterns(pats)
    bases, offsets = _ragged(n, motif, 20250919 + 11 * k + slide)
    lens = np.diff(offsets)
    assert lens.min() < 200 and (lens < 1000).sum() >= 20 and (lens > 20000).sum() >= 200 and lens.max() > 40000
    seq2, inv, desc = seqio.pack_reads_host(bases, offsets)
    assert (desc["flags"] & 1).any()                             # reads with N in the batch
    sc.upload_packed(0, seq2, inv, desc)
    cutoff, min_len = 0.5, 1000                                  # (minSeqLength 1000: the short reads take part in step 1, most long ones pass)
    prm = _params(motif, slide, FULL | (hiplib.F_STORE_RAW if raw else 0), cutoff, min_len)
    sc.scan(0, prm)
    sc.sync()
    res = sc.results(0).copy()
    sums, win_off = sc.window_sums(0)
    rows = sc.window_raw(0)[0] if raw else None
    # every read against the float64 pipeline (what allsteps.py:310-311 computes), all reads
    out, ck = oracle_c.batch_ck(bases, offsets, pats, len(motif), 1000, min_len, cutoff, 100, slide, 100, 20000,
                                both_tails=False, threads=oracle_c.usable_cores(), want_raw=raw)
    assert np.array_equal(res["pass"], out[:, 0])
    p = res["pass"].astype(bool)
    assert 0.5 * n < p.sum() < n
    assert np.array_equal(res["tail"][p], out[p, 1])
    best = np.where(res["tail"] == 0, res["best_start"], res["best_end"])
    assert np.array_equal(best[p], out[p, 3])
    assert np.array_equal(res["n_win"][p], out[p, 4])
    differ = np.nonzero(res["bkp"][p] != out[p, 5])[0]
    assert len(differ) == 0, (differ[:10], res["bkp"][p][differ[:10]], out[p, 5][differ[:10]])
    nw = np.array([hiplib.window_count(int(x), 100, slide, 100, 20000) for x in lens])
    assert np.array_equal(np.diff(win_off), nw)                  # (the layout holds windows for every read, passing or not)
    # EVERY window of every passing read (round 4): per-read checksums of the sums (and of the raw rows) against oracle.c's
    assert np.array_equal(oracle_c.checksums(sums, win_off)[p], ck[p, 0])
    if raw:
        assert np.array_equal(oracle_c.checksums(rows.reshape(-1), win_off * len(pats))[p], ck[p, 1])
    # window by window: every 50th read, plus the shortest and the longest passing ones
    idx = set(range(0, n, 50)) | {int(np.nonzero(p)[0][np.argmin(lens[p])]), int(np.argmax(lens))}
    checked = 0
    for i in sorted(idx):
        if not p[i]:
            continue
        seq = bytes(bases[offsets[i]:offsets[i + 1]]).decode()
        s_c, raw_c = oracle_c.window_counts(seq, TAILS[int(res["tail"][i])], pats, 100, slide, 100, 20000)
        assert np.array_equal(sums[win_off[i]:win_off[i + 1]], s_c), i
        if raw:
            assert np.array_equal(rows[win_off[i]:win_off[i + 1]], raw_c), i
        checked += 1
    assert checked >= 50


def _write_fastq(path, bases, offsets):
    with open(path, "wb") as h:
        for i in range(len(offsets) - 1):
            s = bases[offsets[i]:offsets[i + 1]].tobytes()
            h.write(b"@r%d some text\n" % i + s + b"\n+\n" + b"I" * len(s) + b"\n")


def test_engine_pool_four_contexts_ragged_file(tmp_path):
    """Four contexts on device 0 pull batches of a ragged file from one queue: every batch comes back in file order with the
    rows a single context computes; all contexts upload from the pinned pool the first one allocated (asynchronously: the
    library's registry of pinned buffers is process-wide), and that pool is allocated once, not once per file."""
    motif, k, slide = "CCCTAA", 4, 6
    pats = orc.kmer_table(motif, k)
    bases, offsets = _ragged(2500, motif, 77)
    fq = str(tmp_path / "ragged.fastq")
    _write_fastq(fq, bases, offsets)
    prm = _params(motif, slide, hiplib.F_STEP1 | hiplib.F_WINDOWS | hiplib.F_BINSEG)
    max_bases = 3 << 20                                          # ~ a dozen batches

    def run(engines, times=1):
        ids, rows = [], []
        for _ in range(times):
            ids, rows = [], []
            pool = batch.EnginePool(engines, pats)               # (a new pool per file, as main.process_file_multi makes one)
            for pb, res, _s, _r, _w in pool.scan_file(fq, prm, max_bases=max_bases):
                ids += pb.ids
                rows.append(res.copy())
        return ids, np.concatenate(rows)

    one = [hiplib.HipScanner(0)]
    four = [hiplib.HipScanner(0) for _ in range(4)]
    try:
        ids1, r1 = run(one)
        assert ids1 == [f"r{i}" for i in range(2500)]
        ids4, r4 = run(four)
        pinned_after_first = len(four[0]._pinned)
        assert 0 < pinned_after_first <= 3 * (len(four) + 2) and all(len(getattr(e, "_pinned", {})) == 0 for e in four[1:])
        assert ids4 == ids1
        for f in ("pass", "tail", "best_start", "best_end", "n_win", "bkp"):
            assert np.array_equal(r4[f], r1[f]), f
        ids4b, r4b = run(four, times=3)                          # three more files through the same contexts
        assert ids4b == ids1 and np.array_equal(r4b["bkp"], r1["bkp"])
        # the staging pool is reused, not re-allocated per file: at most its len(engines) + 2 sets (3 pinned arrays each) exist,
        # however many files went through (sets are pinned on demand, so a later file may add the ones the first never needed)
        assert pinned_after_first <= len(four[0]._pinned) <= 3 * (len(four) + 2)
    finally:
        for e in one + four:
            e.close()


def test_tables_scanned_concurrently_equal_back_to_back(sc, monkeypatch):
    """`--telophrase 4 5 6`: batch.scan_jobs gives every table beyond the first to a helper context that borrows the resident
    batch (tps_batch_share) and launches all of them at once.  Row for row, window for window and raw count for raw count the
    answers are those of the same tables scanned back to back on the one context -- batch after batch (the helpers re-borrow
    a slot the owner has re-uploaded, also after it grew), and against the C oracle for the k = 6 table."""
    motif, slide = "CCCTAA", 6
    jobs = [batch.Job(orc.kmer_table(motif, k), _params(motif, slide, hiplib.F_STEP1 | hiplib.F_WINDOWS | hiplib.F_BINSEG),
                      want_sums=True, want_raw=(k != 5)) for k in (4, 5, 6)]
    for n, seed in ((600, 5), (1500, 6), (300, 7)):              # the slot grows, then shrinks
        bases, offsets = _ragged(n, motif, seed)
        recs = type("B", (), {"bases": bases, "offsets": offsets})()     # (goes up as ASCII: batch.upload_batch)
        monkeypatch.setattr(batch, "SEQUENTIAL_TABLES", True)
        seq = batch.scan_jobs(sc, recs, jobs)
        monkeypatch.setattr(batch, "SEQUENTIAL_TABLES", False)
        con = batch.scan_jobs(sc, recs, jobs)
        assert len(sc._helpers) == 2
        for j, ((r1, s1, w1, o1), (r2, s2, w2, o2)) in enumerate(zip(seq, con)):
            for f in r1.dtype.names:
                assert np.array_equal(r1[f], r2[f]), (n, j, f)
            assert np.array_equal(o1, o2) and (w1 is None) == (w2 is None), (n, j)
            keep = np.repeat(r1["pass"].astype(bool), np.diff(o1))       # (windows of reads that fail the filter are never written)
            assert keep.sum() > 0.5 * len(keep) and np.array_equal(s1[keep], s2[keep]), (n, j)
            assert w1 is None or np.array_equal(w1[keep], w2[keep]), (n, j)
        res = con[2][0]
        out, done, _ = oracle_c.batch(bases, offsets, jobs[2].patterns, len(motif), 1000, 1000, 0.5, 100, slide, 100, 20000,
                                      both_tails=False, threads=oracle_c.usable_cores())
        p = res["pass"].astype(bool)
        assert np.array_equal(res["pass"], out[:, 0]) and np.array_equal(res["bkp"][p], out[p, 5])


def test_share_refuses_what_cannot_work(sc):
    other = hiplib.HipScanner(0)
    try:
        with pytest.raises(hiplib.TopsicleHipError, match="another context"):
            sc.share(0, sc, 1)
        with pytest.raises(hiplib.TopsicleHipError, match="no batch"):
            other.share(0, sc, hiplib.MAX_SLOTS - 1)
    finally:
        other.close()


def test_compressed_inputs_scan_like_the_plain_file(tmp_path, monkeypatch):
    """The same ragged reads as plain FASTQ, as ordinary gzip (inflated by the thread team) and as BGZF, through
    batch.EnginePool.scan_file with two contexts: compressed text is packed from reference-counted windows of inflated text
    (several per file here), batches arrive in file order, and ids and result rows are those of the plain file."""
    from topsicle_amd import e2e
    motif, k, slide = "CCCTAA", 4, 6
    pats = orc.kmer_table(motif, k)
    bases, offsets = _ragged(3000, motif, 123)
    fq = str(tmp_path / "r.fastq")
    _write_fastq(fq, bases, offsets)
    import gzip
    import shutil
    gz = str(tmp_path / "g" / "r.fastq.gz")
    bg = str(tmp_path / "b" / "r.fastq.gz")
    os.makedirs(os.path.dirname(gz)); os.makedirs(os.path.dirname(bg))
    with open(fq, "rb") as src, gzip.open(gz, "wb", compresslevel=1) as dst:
        shutil.copyfileobj(src, dst)
    e2e.write_bgzf(bg, fq)
    fa = str(tmp_path / "r.fasta")                              # the same reads as two-line FASTA: the same packed decoder
    with open(fa, "wb") as h:
        for i in range(len(offsets) - 1):
            h.write(b">r%d some text\n" % i + bases[offsets[i]:offsets[i + 1]].tobytes() + b"\n")
    assert os.path.getsize(gz) > (1 << 20)                      # (large enough for the parallel inflater)
    seqio.io_option("bgzf_group", 6 << 20)       # windows of ~6 MB of text: several refills per file
    prm = _params(motif, slide, hiplib.F_STEP1 | hiplib.F_WINDOWS | hiplib.F_BINSEG)
    engines = [hiplib.HipScanner(0), hiplib.HipScanner(0)]
    try:
        got = {}
        for tag, path in (("plain", fq), ("gz", gz), ("bgzf", bg), ("fasta", fa)):
            ids, rows = [], []
            for pb, res, _s, _r, _w in batch.EnginePool(engines, pats).scan_file(path, prm, max_bases=4 << 20):
                ids += pb.ids
                rows.append(res.copy())
            got[tag] = (ids, np.concatenate(rows))
        assert got["plain"][0] == [f"r{i}" for i in range(3000)]
        for tag in ("gz", "bgzf", "fasta"):
            assert got[tag][0] == got["plain"][0], tag
            for f in ("pass", "tail", "best_start", "best_end", "n_win", "bkp"):
                assert np.array_equal(got[tag][1][f], got["plain"][1][f]), (tag, f)
    finally:
        for e in engines:
            e.close()


def test_raw_rows_straight_to_a_file(sc, tmp_path):
    """tps_batch_raw_to_fd (ABI 4): the rows of selected reads, device -> pinned pieces -> pwritev, equal to the rows
    tps_batch_window_raw downloads -- all passing reads (runs that merge, > 32 MB: several pieces), every third read (runs with
    gaps below and above the skip threshold), a sparse handful, nothing; the CRC equals zlib's over the same bytes; two contexts
    write into one file at once (helper context sharing the batch)."""
    import zlib
    motif, k = "CCCTAA", 5
    pats = orc.kmer_table(motif, k)
    sc.set_patterns(pats)
    bases, offsets = _ragged(3000, motif, 77)
    seq2, inv, desc = seqio.pack_reads_host(bases, offsets)
    sc.upload_packed(6, seq2, inv, desc)
    prm = _params(motif, 6, flags=FULL | hiplib.F_STORE_RAW)
    sc.scan(6, prm)
    sc.sync()
    res = sc.results(6)
    raw, win_off = sc.window_raw(6)
    passing = np.nonzero(res["pass"])[0]
    assert 0 < len(passing) < len(res)
    P = len(pats)
    fd = os.open(tmp_path / "rows.bin", os.O_RDWR | os.O_CREAT | os.O_TRUNC)
    try:
        at = 4096
        for sel in (passing, passing[::3], passing[[5, 400, 401, 2000 % len(passing)]] if len(passing) > 2001 else passing[:3], passing[:0],
                    np.arange(len(res))):
            sel = np.unique(sel)
            want = b"".join(raw[win_off[i]:win_off[i + 1]].tobytes() for i in sel)
            got_n, crc = sc.raw_to_fd(6, sel, fd, at)
            assert got_n == len(want)
            assert os.pread(fd, len(want), at) == want
            assert crc == zlib.crc32(want)
            at += len(want) + 17
        assert len(b"".join(raw[win_off[i]:win_off[i + 1]].tobytes() for i in passing)) > (40 << 20)      # several 32 MB pieces were needed
        # a second table on a helper context, same resident batch, both writing into the one file at the same time
        import threading
        h = sc.helper(0)
        h.share(6, sc, 6)
        pats6 = orc.kmer_table(motif, 6)
        h.set_patterns(pats6)
        h.scan(6, prm)
        h.sync()
        raw6, wo6 = h.window_raw(6)
        a_off, b_off = at, at + int(sum(win_off[i + 1] - win_off[i] for i in passing)) * P + 4096
        out = {}
        t = threading.Thread(target=lambda: out.__setitem__("h", h.raw_to_fd(6, passing, fd, b_off)))
        t.start()
        out["s"] = sc.raw_to_fd(6, passing, fd, a_off)
        t.join()
        want_s = b"".join(raw[win_off[i]:win_off[i + 1]].tobytes() for i in passing)
        want_h = b"".join(raw6[wo6[i]:wo6[i + 1]].tobytes() for i in passing)
        assert os.pread(fd, len(want_s), a_off) == want_s and out["s"] == (len(want_s), zlib.crc32(want_s))
        assert os.pread(fd, len(want_h), b_off) == want_h and out["h"] == (len(want_h), zlib.crc32(want_h))
    finally:
        os.close(fd)
    with pytest.raises(hiplib.TopsicleHipError):
        sc.raw_to_fd(6, np.array([3, 2]), 1, 0)                # not ascending


def test_cli_npz_on_gpu_equals_the_per_read_csv_rows(tmp_path, gold_dir, monkeypatch):
    """--rawcountformat npz through the real CLI: many batches on two contexts, three k -- the archive verifies (zip CRCs), its
    rows equal the reference's rawCountPattern matrices (tests/golden/demo_windows.npz) and the oracle's on synthetic reads."""
    import json
    import shutil
    import zipfile
    from topsicle_amd import main as cli
    d = tmp_path / "in"
    d.mkdir()
    fq = d / "demo.fastq.gz"
    shutil.copyfile(os.path.join(gold_dir, "demo_col0.fastq.gz"), fq)
    out = tmp_path / "out"
    cli.main(["-i", str(fq), "-o", str(out), "--pattern", "CCCTAAA", "--slide", "6", "--rawcountpattern", "--rawcountformat", "npz"])
    meta = json.load(open(os.path.join(gold_dir, "demo_windows.json")))
    arrs = np.load(os.path.join(gold_dir, "demo_windows.npz"))
    with zipfile.ZipFile(out / "rawcount_5_demo.fastq.npz") as zf:
        assert zf.testzip() is None
    z = np.load(out / "rawcount_5_demo.fastq.npz")
    ids = z["read_id"].tolist()
    assert len(ids) == 17
    for j, r in enumerate(meta["reads"]):
        if f"counts_{j}" in arrs:
            i = ids.index(r["id"])
            assert np.array_equal(z["counts"][z["win_off"][i]:z["win_off"][i + 1]], arrs[f"counts_{j}"])
            assert str(z["tail"][i]) == r["tail"]
    # synthetic, several batches, k = 4, 5, 6
    monkeypatch.setattr(batch, "BATCH_BASES", 4 << 20)
    bases, offsets, _truth = synth.make_reads(900, 12000, "CCCTAA", seed=5)
    fq2 = tmp_path / "s.fastq"
    _write_fastq(fq2, bases, offsets)
    out2 = tmp_path / "out2"
    cli.main(["-i", str(fq2), "-o", str(out2), "--pattern", "CCCTAA", "--telophrase", "4", "5", "6", "--rawcountpattern", "--rawcountformat", "npz"])
    for k in (4, 5, 6):
        with zipfile.ZipFile(out2 / f"rawcount_{k}_s.npz") as zf:
            assert zf.testzip() is None
        z = np.load(out2 / f"rawcount_{k}_s.npz")
        pats = orc.kmer_table("CCCTAA", k)
        ids = z["read_id"].tolist()
        nums = [int("".join(ch for ch in r if ch.isdigit())) for r in ids]
        assert len(ids) > 700 and nums == sorted(nums)                        # file order, although the batches finish out of order
        for i in range(0, len(ids), 97):
            rid = ids[i]
            ridx = int("".join(ch for ch in rid if ch.isdigit()))
            seq = bases[offsets[ridx]:offsets[ridx + 1]].tobytes().decode()
            want = orc.window_count_matrix(seq, str(z["tail"][i]), pats, 100, 6, 100, 20000)[1]
            assert np.array_equal(z["counts"][z["win_off"][i]:z["win_off"][i + 1]], want), (k, rid)


def test_one_file_cut_into_shards_on_four_contexts(tmp_path):
    """One plain FASTQ file, 3 byte ranges with a reader team each, four contexts ("2 GPUs x 2 contexts") on this box's GPU: rows
    equal the one-reader run's, in file order, every seam closed; the heads-mode (two-pass) route through the shards too."""
    bases, offsets = _ragged(2600, "CCCTAA", 41)
    fq = tmp_path / "big.fastq"
    _write_fastq(fq, bases, offsets)
    motif = "CCCTAA"
    pats = orc.kmer_table(motif, 4)
    prm = _params(motif, 6, flags=hiplib.F_STEP1 | hiplib.F_WINDOWS | hiplib.F_BINSEG)
    engines = [hiplib.HipScanner(0) for _ in range(4)]
    try:
        def rows(shards, two_pass):
            ep = batch.EnginePool(engines, pats, two_pass=two_pass)
            out = []
            for pb, outs in ep.scan_file_jobs(str(fq), [batch.Job(pats, prm)], max_bases=4 << 20, shards=shards, shard_min_bytes=1 << 20):
                res = outs[0][0]
                for i in range(len(res)):
                    out.append((pb.read_id(i), int(res["pass"][i]), int(res["tail"][i]), int(res["best_start"][i]), int(res["best_end"][i]),
                                int(res["n_win"][i]), int(res["bkp"][i])))
            return out, ep.stats
        one, _ = rows(1, "off")
        cut, st = rows(3, "off")
        assert st.get("shards") == 3 and len(one) == 2600
        assert cut == one
        cut2, st2 = rows(3, "on")
        assert st2.get("shards") == 3 and st2["heads_batches"] > 0
        assert cut2 == one
    finally:
        for e in engines:
            e.close()


def test_cli_slide_10_with_raw_rows_and_two_pass_equals_the_generic_kernel(tmp_path, monkeypatch):
    """`--slide 10 --telophrase 4 6 --rawcountpattern` through the real CLI (several batches, two contexts, the two-pass reader, rows
    device -> file): the strided scans (tps::stride_base: the slide-5 kernels, every second window) against the same run on the generic
    kernel (TOPSICLE_HIP_DEBUG=no_stride=1) -- CSV rows, archives' rows and filtered files equal."""
    import zipfile
    from topsicle_amd import main as cli
    monkeypatch.setattr(batch, "BATCH_BASES", 3 << 20)
    bases, offsets, _truth = synth.make_ragged_reads(700, "CCCTAA", 91, len_mu=9.4, len_sigma=0.5, max_len=24000, telomeric_fraction=0.6)
    fq = tmp_path / "s.fastq"
    _write_fastq(fq, bases, offsets)
    outs = {}
    for mode in ("strided", "generic"):
        if mode == "generic":
            monkeypatch.setenv("TOPSICLE_HIP_DEBUG", "no_stride=1")
        else:
            monkeypatch.delenv("TOPSICLE_HIP_DEBUG", raising=False)
        out = tmp_path / mode
        cli.main(["-i", str(fq), "-o", str(out), "--pattern", "CCCTAA", "--telophrase", "4", "6", "--slide", "10", "--rawcountpattern", "--rawcountformat", "npz",
                  "--twopass", "on"])
        outs[mode] = out
    a, b = outs["strided"], outs["generic"]
    rows_a = open(a / "telolengths_all.csv").read()
    assert rows_a == open(b / "telolengths_all.csv").read() and rows_a.count("\n") > 300
    for k in (4, 6):
        for o in (a, b):
            with zipfile.ZipFile(o / f"rawcount_{k}_s.npz") as zf:
                assert zf.testzip() is None
        za, zb = np.load(a / f"rawcount_{k}_s.npz"), np.load(b / f"rawcount_{k}_s.npz")
        for f in ("read_id", "tail", "win_off", "counts"):
            assert np.array_equal(za[f], zb[f]), (k, f)
        assert len(za["read_id"]) > 100
    for f in sorted(os.listdir(a)):
        if "_trc_over_" in f:
            assert open(a / f).read() == open(b / f).read(), f
