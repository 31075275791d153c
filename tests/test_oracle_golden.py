"""Pins the CPU oracle (oracle/topsicle_oracle.py) against fixtures produced by the reference's
own code (oracle/gen_golden.py) and against the reference's shipped demo results."""
import csv
import json
import os
import re

import numpy as np
import pytest

import topsicle_oracle as orc


def test_pattern_tables(gold_dir):
    for c in json.load(open(os.path.join(gold_dir, "patterns.json"))):
        if c["scramble"] is not None:
            assert orc.kmers_of_repeat(c["motif"], c["k"]) == c["scramble"], c
        assert orc.kmer_table(c["motif"], c["k"]) == c["search"], c


def test_demo_log_pattern_line(gold_dir):
    log = json.load(open(os.path.join(gold_dir, "demo_run_log.json")))
    assert str(orc.kmer_table(log["pattern"], log["telophrase"])) == log["patterns_line"]


def test_nonoverlap_count_is_finditer():
    rng = np.random.default_rng(7)
    pats = ["CTAAC", "CCC", "AA", "ACCCTA", "CCTAAC", "GATTG", "A", "ATAT", "AACC"]
    for _ in range(300):
        n = int(rng.integers(0, 120))
        text = "".join("ACGTN"[i] for i in rng.choice(5, n, p=[.3, .35, .1, .2, .05]))
        for p in pats:
            assert orc.nonoverlap_count(p, text) == len(list(re.finditer(p, text)))
    assert orc.nonoverlap_count("CTAAC", "CTAACTAACTAAC") == 2
    assert orc.nonoverlap_count("CCC", "CCCCCCC") == 2


def test_demo_step1(gold_dir, demo_records):
    for case in json.load(open(os.path.join(gold_dir, "demo_step1.json"))):
        rows = orc.step1(demo_records, case["motif"], case["k"], case["min_len"], case["cutoff"])
        assert rows == case["rows"], (case["motif"], case["k"], case["cutoff"])


def test_demo_windows_and_boundaries(gold_dir, demo_windows, demo_records):
    meta, arrs = demo_windows
    seqs = dict(demo_records)
    pats = meta["patterns"]
    gold_csv = list(csv.reader(open(os.path.join(gold_dir, "demo_telolengths_all.csv"))))[1:]
    n17 = 0
    for i, r in enumerate(meta["reads"]):
        s = r.get("slide", meta["slide"])
        key = r.get("key")
        want = arrs[f"{key.replace('s7_', 's7_counts_')}" if key else f"counts_{i}"]
        y = arrs[f"{key.replace('s7_', 's7_y_')}" if key else f"y_{i}"]
        _, counts = orc.window_count_matrix(seqs[r["id"]], r["tail"], pats, meta["W"], s,
                                            meta["trimfirst"], meta["maxlengthtelo"])
        assert np.array_equal(counts, want)
        sums = orc.window_sums(counts)
        assert np.array_equal(sums / len(pats), y)          # bit-identical y
        b = orc.boundary_from_sums(sums, len(pats), s, meta["trimfirst"], meta["maxlengthtelo"])
        assert b == r["boundary"]
        if not key:
            assert gold_csv[i][3] == r["id"] and int(gold_csv[i][4]) == b
            assert f"{r['trc']:.3f}" == gold_csv[i][2]
            n17 += 1
    assert n17 == 17


def test_synthetic_cases(synth_cases):
    meta, arrs = synth_cases
    for ci, c in enumerate(meta):
        pats = orc.kmer_table(c["motif"], c["k"])
        assert pats == c["patterns"]
        rows = orc.step1([(c["id"], c["seq"])], c["motif"], c["k"], 0, -1.0, c["no_bp"])
        assert (rows[0][1:] if rows else None) == c["step1"], c["name"]
        for tail in c["tails"]:
            starts, counts = orc.window_count_matrix(c["seq"], tail, pats, c["W"], c["s"], c["t"], c["M"])
            assert np.array_equal(np.asarray(starts, np.int32), arrs[f"starts_{ci}_{tail}"]), c["name"]
            assert np.array_equal(counts.reshape(-1, len(pats)), arrs[f"counts_{ci}_{tail}"]), c["name"]
            y = arrs[f"y_{ci}_{tail}"]
            if counts.shape[0]:
                sums = orc.window_sums(counts)
                if y.size:
                    assert np.array_equal(sums / len(pats), y), c["name"]
                got = orc.boundary_from_sums(sums, len(pats), c["s"], c["t"], min(c["M"], len(c["seq"])))
                assert got == c["boundary"][tail], c["name"]
                if c["binseg_error"][tail]:
                    assert got is None


def exact_split_value(sums, b):
    from fractions import Fraction
    s = [int(v) for v in sums]
    left = sum(s[:b])
    return Fraction(left * left, b) + Fraction((sum(s) - left) ** 2, len(s) - b)


def test_exact_binseg_matches_float_on_goldens(synth_cases, demo_windows):
    """The integer tie rule (what the GPU kernel implements) agrees with the numpy-float
    formulation on every golden vector; a disagreement must be a tie that float64 cannot
    resolve (exact gains equal to 1e-12 relative), e.g. the constant polyC signal."""
    n = ties = 0
    for meta, arrs in (synth_cases, demo_windows):
        for key in arrs.files:
            if "counts" not in key:
                continue
            sums = arrs[key].astype(np.int64).sum(axis=1)
            P = arrs[key].shape[1]
            bf, _ = orc.binseg_l2_numpy(sums / P)
            be = orc.binseg_l2_exact(sums)
            if bf != be:
                vf, ve = exact_split_value(sums, bf), exact_split_value(sums, be)
                assert ve >= vf and float(ve - vf) <= 1e-12 * float(ve), key
                ties += 1
            n += 1
    assert n > 60 and ties <= 3


def test_quadratic_vertex_demo(gold_dir):
    """Host summary known-answer: medians and asymptotic TRC of the demo log."""
    rows = list(csv.reader(open(os.path.join(gold_dir, "demo_telolengths_all.csv"))))[1:]
    trc = [float(r[2]) for r in rows]
    telo = [float(r[4]) for r in rows]
    log = json.load(open(os.path.join(gold_dir, "demo_run_log.json")))
    assert f"{np.median(telo):.2f}" in log["median_line"]
    vx, _, _ = orc.quadratic_vertex(trc, telo, 0.7, np.median(trc))
    assert f"{vx:.3f}" in log["asymptotic_line"]
    kept = [t for c, t in zip(trc, telo) if c >= vx]
    assert f"{np.median(kept):.2f}" in log["filtered_line"]
