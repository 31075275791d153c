"""Pins oracle/oracle.c (the C restatement used for the CPU baseline and for large-size checks)
against the Python oracle, the reference-generated goldens and numpy's float64 variance."""
import numpy as np
import pytest

import oracle_c
import topsicle_oracle as orc


def test_c_counts_on_goldens(synth_cases):
    meta, arrs = synth_cases
    for ci, c in enumerate(meta):
        pats = c["patterns"]
        cs, ce = oracle_c.trc_counts(c["seq"], pats, c["no_bp"])
        ws, we = orc.trc_counts(c["seq"], pats, c["no_bp"])
        assert (cs, ce) == (ws, we), c["name"]
        for tail in c["tails"]:
            want = arrs[f"counts_{ci}_{tail}"].astype(np.int64)
            sums, raw = oracle_c.window_counts(c["seq"], tail, pats, c["W"], c["s"], c["t"], c["M"])
            assert raw.shape[0] == want.shape[0], c["name"]
            if want.shape[0]:
                assert np.array_equal(raw, want) and np.array_equal(sums, want.sum(axis=1)), c["name"]
                b, g = oracle_c.binseg_l2(sums, len(pats))
                bp, gp = orc.binseg_l2_numpy(sums / len(pats))
                assert b == bp and (b is None or g == gp), c["name"]      # bit-identical float64 gain


def test_c_binseg_is_numpy_bit_for_bit():
    rng = np.random.default_rng(11)
    for n in list(range(0, 40)) + [127, 128, 129, 255, 256, 257, 1000, 2467, 2829, 3301]:
        for rep in range(3):
            s = rng.integers(12, 400, n)
            if n > 30 and rep:
                s[: n // (rep + 1)] += 100
            b, g = oracle_c.binseg_l2(s, 12)
            bp, gp = orc.binseg_l2_numpy(s / 12)
            assert b == bp, n
            if b is not None:
                assert g == gp, (n, g, gp)
    y = rng.normal(size=777)
    assert oracle_c.binseg_l2_y(y) == orc.binseg_l2_numpy(y)


def test_c_demo_pipeline(demo_records, gold_dir):
    import csv, os
    from topsicle_amd import hiplib
    pats = orc.kmer_table("CCCTAAA", 5)
    bases, offsets = hiplib.pack_reads([s for _, s in demo_records])
    out, done, _ = oracle_c.batch(bases, offsets, pats, 7, 1000, 9000, 0.7, 100, 6, 100, 20000, both_tails=True, threads=4)
    assert done == len(demo_records)
    gold = list(csv.reader(open(os.path.join(gold_dir, "demo_telolengths_all.csv"))))[1:]
    got = [(demo_records[i][0], int(out[i, 6]), f"{out[i, 3] / (1000 / 7):.3f}") for i in range(len(demo_records)) if out[i, 0]]
    assert got == [(g[3], int(g[4]), g[2]) for g in gold]
