"""Exact ties of the change-point: where two or more candidates lie within float64 rounding noise
of the best gain, ruptures' answer is decided by that noise.  The kernels decide such reads with an exact integer tournament
AND flag them (TPS_RES_TIE); the host then repeats ruptures' float64 arithmetic on the read's S_w, so that what the package
reports is what `rpt.Binseg(model="l2").fit(y).predict(n_bkps=1)` reports -- also for degenerate signals.

CPU part: through the emulation.  GPU part (-m gpu): the same through the C ABI, the standalone entry point and the CLI."""
import os

import numpy as np
import pytest

import emu_driver as emu
import topsicle_oracle as orc
from emu_engine import EmuEngine
from topsicle_amd import allsteps, batch, hiplib


def test_host_float64_binseg_is_the_restated_ruptures_arithmetic():
    rng = np.random.default_rng(3)
    cases = [np.tile([12.0, 13.0], 200), np.full(300, 12.0), np.tile([1.0, 2.0, 3.0], 50)]
    cases += [rng.integers(12, 40, rng.integers(7, 400)).astype(np.float64) for _ in range(100)]
    for y in cases:
        want, _ = orc.binseg_l2_numpy(y / 12)
        assert hiplib.binseg_l2_float64(y / 12) == (-1 if want is None else want)
    assert hiplib.binseg_l2_float64(np.tile([12.0, 13.0], 200) / 12) != orc.binseg_l2_exact(np.tile([12, 13], 200))      # the case that needs all this


def _tie_signals():
    return [np.tile([12, 13], 200), np.full(301, 12), np.full(64, 14), np.tile([12, 12, 13], 100), np.arange(100) % 2 + 12]


def _check_standalone(engine):
    for s in _tie_signals():
        s = np.asarray(s, np.int32)
        bkp, _ = engine.binseg_l2(s, np.array([0, len(s)], np.int64), 12)
        want, _ = orc.binseg_l2_numpy(s / 12)
        assert bkp[0] == want, (s[:6], len(s))


def _check_fused(engine):
    """Reads whose window sums are constant (no k-mer of the table anywhere: S_w = P in every window) through the batch driver:
    the kernel flags them, the driver hands back ruptures' answer."""
    pats = orc.kmer_table("CCCTAA", 4)
    engine.set_patterns(pats)
    seqs = ["C" * 900, "A" * 2500, "G" * 1234]
    prm = hiplib.make_params(no_bp=1000, min_len=0, min_count=-1, window=100, slide=6, trimfirst=100, maxlen=20000,
                             flags=hiplib.F_STEP1 | hiplib.F_WINDOWS | hiplib.F_BINSEG)
    res, _s, _r, _w = batch.scan_records(engine, [type("R", (), {"seq": s})() for s in seqs], prm)
    assert res["pass"].all() and (res["flags"] & hiplib.RES_TIE).all()
    for i, s in enumerate(seqs):
        n_win = hiplib.window_count(len(s), 100, 6, 100, 20000)
        want, _ = orc.binseg_l2_numpy(np.full(n_win, 12.0) / 12)
        assert res["n_win"][i] == n_win and res["bkp"][i] == want
    # ordinary reads are not flagged
    from topsicle_amd import synth
    b, o, _ = synth.make_reads(8, 3000, "CCCTAA", seed=5, tract_min=300, tract_max=1500)
    res2, _s, _r, _w = batch.scan_records(engine, [type("R", (), {"seq": s})() for s in synth.split_reads(b, o)], prm)
    assert not (res2["flags"] & hiplib.RES_TIE).any()


def test_ties_through_the_emulation():
    e = EmuEngine()
    _check_standalone(e)
    _check_fused(e)


@pytest.mark.gpu
def test_ties_on_the_gpu():
    with hiplib.HipScanner(0) as sc:
        _check_standalone(sc)
        _check_fused(sc)


@pytest.mark.gpu
def test_ties_through_the_cli_on_gpu(tmp_path):
    """A file of degenerate reads (poly-C with a telomeric head so that they pass the TRC filter) through `topsicle`: the reported
    lengths are the float64 restatement's."""
    import csv
    from topsicle_amd import main as cli
    motif = "CCCTAA"
    head = (motif * 200)[:1100]
    seqs = {f"t{i}": head + "C" * n for i, n in enumerate((3000, 5000, 9001))}
    d = tmp_path / "in"
    d.mkdir()
    with open(d / "ties.fastq", "w") as h:
        for rid, s in seqs.items():
            h.write(f"@{rid}\n{s}\n+\n{'I' * len(s)}\n")
    out = tmp_path / "out"
    args = cli.build_parser().parse_args(["-i", str(d / "ties.fastq"), "-o", str(out), "--pattern", motif, "--minSeqLength", "2000", "--trimfirst", "1200"])
    cli.tprint.logfile = cli.get_log_path(args)
    cli.analysis_run(args)
    rows = list(csv.reader(open(out / "telolengths_all.csv")))[1:]
    assert [r[3] for r in rows] == list(seqs)
    pats = orc.kmer_table(motif, 4)
    for r in rows:
        want = orc.step2(seqs[r[3]], "forward", pats, 100, 6, 1200, 20000)      # float64 restatement on the constant signal
        assert int(r[4]) == want
