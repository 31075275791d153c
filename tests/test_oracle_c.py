"""Pins oracle/oracle.c (the C restatement used for the CPU baseline and for large-size checks)
against the Python oracle, the reference-generated goldens and numpy's float64 variance."""
import numpy as np
import pytest

import oracle_c
import topsicle_oracle as orc
from topsicle_amd import synth


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


def test_batch_checksums_equal_the_vectorised_ones():
    """oracle.c's per-read checksums of the window sums / raw rows (orc_batch_ck) = oracle_c.checksums over the same vectors laid out
    read after read -- what the -m gpu tests compare the kernels' outputs with, every window of every read."""
    motif, k = "CCCTAA", 6
    pats = orc.kmer_table(motif, k)
    bases, offsets, _ = synth.make_reads(24, 10500, motif, seed=5)
    out, ck = oracle_c.batch_ck(bases, offsets, pats, 6, 1000, 9000, 0.5, 100, 6, 100, 20000, threads=3, want_raw=True)
    sums, rows, wo = [], [], [0]
    for i in range(24):
        s_c = np.zeros(0, np.int32)
        r_c = np.zeros((0, len(pats)), np.uint8)
        if out[i, 0]:
            s_c, r_c = oracle_c.window_counts(bytes(bases[offsets[i]:offsets[i + 1]]).decode(), ["forward", "reverse"][out[i, 1]], pats, 100, 6, 100, 20000)
        sums.append(s_c)
        rows.append(r_c.reshape(-1))
        wo.append(wo[-1] + len(s_c))
    wo = np.array(wo)
    assert out[:, 0].sum() >= 12
    assert np.array_equal(oracle_c.checksums(np.concatenate(sums), wo), ck[:, 0])
    assert np.array_equal(oracle_c.checksums(np.concatenate(rows), wo * len(pats)), ck[:, 1])
    s2 = np.concatenate(sums).copy()
    s2[[5, 6]] = s2[[6, 5]] if s2[5] != s2[6] else s2[[5, 6]] + [1, 0]      # a swap (or a change) of two windows is seen
    assert not np.array_equal(oracle_c.checksums(s2, wo), ck[:, 0])
