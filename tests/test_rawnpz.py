"""--rawcountformat npz written device -> file (topsicle_amd/rawnpz.py): the archive np.load reads, member CRCs that verify, rows in
file order although batches finish out of order, reads without a window, --read_check, and no half-written file after a failure.
Runs on the emulated engines (the host-side plumbing is the same; `-m gpu`: tests/test_gpu_pipeline.py drives the real
tps_batch_raw_to_fd)."""
import os
import threading
import zipfile
import zlib

import numpy as np
import pytest

from topsicle_amd import batch, hiplib, main as cli, rawnpz


def run_cli(engines, argv):
    args = cli.build_parser().parse_args(argv)
    cli.tprint.logfile = cli.get_log_path(args)
    cli.analysis_run(args, engines=engines)
    return args


def telomeric_reads(n, seed, lengths):
    rng = np.random.default_rng(seed)
    recs = []
    for i in range(n):
        L = int(lengths[i % len(lengths)])
        tel = ("CCCTAA" * 400)[: min(L, 1500)]
        rest = "".join("ACGT"[c] for c in rng.integers(0, 4, max(L - len(tel), 0)))
        s = tel + rest
        if i % 3 == 1:                                     # telomere at the 3' end: the reverse tail
            s = s[::-1]
        recs.append((f"read{i}", s))
    return recs


def write_fastq(path, recs):
    with open(path, "w") as h:
        for rid, s in recs:
            h.write(f"@{rid}\n{s}\n+\n{'I' * len(s)}\n")


def test_crc32_combine_matches_zlib():
    rng = np.random.default_rng(1)
    for la, lb in [(0, 5), (7, 0), (1, 1), (100, 65537), (4096, 33)]:
        a, b = rng.integers(0, 256, la, dtype=np.uint8).tobytes(), rng.integers(0, 256, lb, dtype=np.uint8).tobytes()
        assert rawnpz._crc32_combine(zlib.crc32(a), zlib.crc32(b), lb) == zlib.crc32(a + b)


def test_archive_layout_is_a_zip_numpy_reads(tmp_path):
    """The hand-made container: np.load, zipfile.testzip (member CRCs), rows at DATA_OFF."""
    class Eng:
        def __init__(self, rows):
            self.rows = rows

        def raw_to_fd(self, slot, reads, fd, off):
            blob = b"".join(self.rows[int(i)].tobytes() for i in reads)
            os.pwrite(fd, blob, off)
            return len(blob), zlib.crc32(blob)
    rng = np.random.default_rng(2)
    pats = ["AAAC", "AACC", "ACCC", "CCCT"]
    w = rawnpz.RawNpzWriter(str(tmp_path / "x.npz"), pats, 6)
    rows = [rng.integers(0, 9, (n, 4), dtype=np.uint8) for n in (5, 0, 17, 3)]
    eng = Eng(rows)
    # batches 1 and 0 finish out of order; batch 2 has nothing
    t = threading.Thread(target=lambda: w.write_batch(1, eng, 0, np.array([2, 3]), 20))
    t.start()
    w.write_batch(0, eng, 0, np.array([0, 1]), 5)
    t.join()
    w.skip_batch(2)
    w.add_reads(["a", "b"], ["forward", "reverse"], [5, 0])
    w.add_reads(["c", "d"], ["forward", "forward"], [17, 3])
    w.finish()
    with zipfile.ZipFile(tmp_path / "x.npz") as z:
        assert z.testzip() is None
        assert z.namelist()[0] == "counts.npy" and set(z.namelist()) == {"counts.npy", "read_id.npy", "tail.npy", "win_off.npy", "pattern.npy", "slide.npy"}
        assert z.getinfo("counts.npy").header_offset == 0
    z = np.load(tmp_path / "x.npz")
    assert z["counts"].shape == (25, 4) and z["counts"].dtype == np.uint8
    assert np.array_equal(z["counts"], np.concatenate(rows))
    assert z["win_off"].tolist() == [0, 5, 5, 22, 25] and z["read_id"].tolist() == ["a", "b", "c", "d"]
    assert z["pattern"].tolist() == pats and int(z["slide"]) == 6 and z["tail"].tolist() == ["forward", "reverse", "forward", "forward"]
    raw = open(tmp_path / "x.npz", "rb").read()
    assert raw[rawnpz.DATA_OFF:rawnpz.DATA_OFF + 100] == np.concatenate(rows).tobytes()[:100]       # (a consumer may map the rows in place)


def test_cli_npz_several_batches_both_tails_reads_without_windows(emu_engine_factory, tmp_path, monkeypatch):
    """Many small batches on two contexts (rows must land in file order), reads shorter than trimfirst + window (they pass,
    have no window: ADVICE r4 -- the round-4 writer raised TypeError on their empty block), three k."""
    monkeypatch.setattr(batch, "BATCH_BASES", 20000)
    recs = telomeric_reads(23, 7, [2500, 1800, 3100, 160, 199, 2000, 205])
    fq = tmp_path / "r.fastq"
    write_fastq(fq, recs)
    out = tmp_path / "o"
    engines = emu_engine_factory()
    run_cli(engines, ["-i", str(fq), "-o", str(out), "--pattern", "CCCTAA", "--telophrase", "4", "5", "--minSeqLength", "100",
                      "--rawcountpattern", "--rawcountformat", "npz", "--maxlengthtelo", "2400", "--cutoff", "0.05"])
    import topsicle_oracle as orc
    seqs = dict(recs)
    for k in (4, 5):
        path = out / f"rawcount_{k}_r.npz"
        with zipfile.ZipFile(path) as zf:
            assert zf.testzip() is None
        z = np.load(path)
        pats = orc.kmer_table("CCCTAA", k)
        ids = z["read_id"].tolist()
        assert ids == [rid for rid, _ in recs]                         # every read is telomeric: all pass, in file order
        n_win = np.diff(z["win_off"])
        assert (n_win == 0).sum() >= 3                                 # the 160- and 199-base reads (and 205 - 100 - 100 + 1 ...)
        for i, rid in enumerate(ids):
            tail = str(z["tail"][i])
            want = orc.window_count_matrix(seqs[rid], tail, pats, 100, 6, 100, 2400)[1]
            got = z["counts"][z["win_off"][i]:z["win_off"][i + 1]]
            assert got.shape == want.shape and np.array_equal(got, want), (k, rid)
    assert not [f for f in os.listdir(out) if f.endswith(".rows")]


def test_cli_npz_read_check_and_two_pass(emu_engine_factory, tmp_path, monkeypatch):
    """--read_check keeps one read's rows; the two-pass route (heads first, then the passing reads as a batch of their own) writes
    the same archive as the one-pass route."""
    monkeypatch.setattr(batch, "BATCH_BASES", 60000)
    rng = np.random.default_rng(11)
    recs = telomeric_reads(9, 3, [5200, 4100, 6000])
    junk = [(f"junk{i}", "".join("ACGT"[c] for c in rng.integers(0, 4, 5000))) for i in range(14)]
    mixed = [x for pair in zip(junk, recs + recs[:5]) for x in pair][:23]
    mixed = [(f"{rid}_{j}", s) for j, (rid, s) in enumerate(mixed)]
    fq = tmp_path / "m.fastq"
    write_fastq(fq, mixed)
    outs = {}
    for mode in ("off", "on"):
        out = tmp_path / f"o_{mode}"
        run_cli(emu_engine_factory(), ["-i", str(fq), "-o", str(out), "--pattern", "CCCTAA", "--telophrase", "4", "--minSeqLength", "1000",
                                       "--rawcountpattern", "--rawcountformat", "npz", "--twopass", mode])
        with zipfile.ZipFile(out / "rawcount_4_m.npz") as zf:
            assert zf.testzip() is None
        outs[mode] = {k: v for k, v in np.load(out / "rawcount_4_m.npz").items()}
    assert set(outs["off"]) == set(outs["on"])
    for k in outs["off"]:
        assert np.array_equal(outs["off"][k], outs["on"][k]), k
    assert len(outs["off"]["read_id"]) > 0 and all("read" in r for r in outs["off"]["read_id"].tolist())
    one = outs["off"]["read_id"].tolist()[2]
    out = tmp_path / "o_check"
    run_cli(emu_engine_factory(), ["-i", str(fq), "-o", str(out), "--pattern", "CCCTAA", "--telophrase", "4", "--minSeqLength", "1000",
                                   "--rawcountpattern", "--rawcountformat", "npz", "--read_check", one])
    z = np.load(out / "rawcount_4_m.npz")
    assert z["read_id"].tolist() == [one]
    lo, hi = outs["off"]["win_off"][2], outs["off"]["win_off"][3]
    assert np.array_equal(z["counts"], outs["off"]["counts"][lo:hi])


def test_no_archive_left_behind_when_a_batch_fails(emu_engine_factory, tmp_path, monkeypatch):
    monkeypatch.setattr(batch, "BATCH_BASES", 20000)
    recs = telomeric_reads(12, 5, [2500, 3000])
    fq = tmp_path / "f.fastq"
    write_fastq(fq, recs)
    engines = emu_engine_factory()
    calls = {"n": 0}
    real = type(engines[0]).raw_to_fd

    def flaky(self, slot, reads, fd, off):
        calls["n"] += 1
        if calls["n"] == 2:
            raise hiplib.TopsicleHipError("injected write failure")
        return real(self, slot, reads, fd, off)
    monkeypatch.setattr(type(engines[0]), "raw_to_fd", flaky)
    out = tmp_path / "o"
    with pytest.raises(hiplib.TopsicleHipError):
        run_cli(engines, ["-i", str(fq), "-o", str(out), "--pattern", "CCCTAA", "--telophrase", "4", "--minSeqLength", "100",
                          "--rawcountpattern", "--rawcountformat", "npz"])
    assert not [f for f in os.listdir(out) if f.startswith("rawcount_")]
