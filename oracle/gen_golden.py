#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/* by running the REFERENCE's own code.

Runs in the build container only (needs /root/reference).  It imports the reference's
Topsicle/allsteps.py unchanged through oracle/ref_import.py and records inputs + outputs of:

  patterns.json        pattern_scramble_telo / patterns_to_search          (allsteps.py:57-125)
  demo_step1.json      patternTRC_count on the demo fastq, several cutoffs (allsteps.py:152-204)
  demo_windows.npz     rawCountPattern matrices, bound_detect's y vectors and boundaries for
                       the 17 demo reads of Topsicle_demo/telolengths_all.csv
                       (allsteps.py:227-338, 359-464)
  synth_cases.json/.npz  the same three calls on small seeded synthetic reads that exercise
                       lower case, non-ACGT letters, self-overlapping k-mers, short reads,
                       odd window/slide/trim values
  cli_<seed>.json      whole runs of the reference's main() on seeded inputs (oracle/cli_cases.py): flags, input files, and the
                       CSV rows / summary log lines / filtered files the run left behind   (main.py:52-154, 156-309)
  demo_col0.fastq.gz, demo_telolengths_all.csv, demo_run_log.json
                       data files of the reference's demo (inputs / its only result goldens)

Boundaries come from the reference's bound_detect driven by the Binseg restatement
(ruptures is not installable here) -- they are pinned by the 17 demo values only.

    python oracle/gen_golden.py            # rewrites tests/golden/
"""
import csv
import hashlib
import json
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")
REF = ref_import.REFERENCE_ROOT
DEMO_FQ = os.path.join(REF, "Topsicle_demo", "data_col0_teloreg_chr",
                       "Col-0-6909_GWHBDNP00000001.1_nano_right.fastq.gz")
DEMO_CSV = os.path.join(REF, "Topsicle_demo", "telolengths_all.csv")


def df_to_matrix(df, n_patterns):
    """rawCountPattern's DataFrame (window-major, then pattern) -> (starts, counts[n_win,P])."""
    cnt = df["count"].to_numpy().astype(np.int32)
    pos = df["position"].to_numpy().astype(np.int64)
    n_win = len(cnt) // n_patterns
    return pos.reshape(n_win, n_patterns)[:, 0].copy(), cnt.reshape(n_win, n_patterns)


def run_step2(ref, path, rid, pats, W, s, t, M, k, tail):
    """bound_detect + rawCountPattern of the reference for one read; returns dict of arrays."""
    ref_import.CAPTURED_Y.clear()
    try:
        bound = ref.bound_detect(path, rid, pats, W, s, t, M, k, tail=tail)
        err = None
    except RuntimeError as e:           # stand-in for ruptures' BadSegmentationParameters
        bound, err = [], str(e)
    y = ref_import.CAPTURED_Y[0].copy() if ref_import.CAPTURED_Y else np.zeros(0)
    df = ref.rawCountPattern(path, rid, pats, W, s, t, k, 0, M, tail=tail)
    starts, counts = df_to_matrix(df, len(pats)) if len(df) else (np.zeros(0, np.int64), np.zeros((0, len(pats)), np.int32))
    return dict(boundary=(bound[0][1] if bound else None), binseg_error=err, y=y, starts=starts, counts=counts)


# ----------------------------------------------------------------------------- patterns
def gen_patterns(ref):
    cases = []
    for motif, ks in [("CCCTAA", [3, 4, 5, 6]), ("AAACCCT", [4, 5, 6, 7]), ("CCCTAAA", [5]),
                      ("TTAGGG", [4]), ("ccctaa", [4]), ("TTAGG", [3]), ("AT", [1, 2]),
                      ("CTCGGTTATGGG", [8, 10]), ("A", [1]), ("TTAGGC", [4, 5])]:
        for k in ks:
            cases.append(dict(motif=motif, k=k,
                              scramble=ref.pattern_scramble_telo(motif, k),
                              search=ref.patterns_to_search(motif, k)))
    cases.append(dict(motif=["aacc", "TTGG"], k=4, scramble=None,
                      search=ref.patterns_to_search(["aacc", "TTGG"], 4)))
    json.dump(cases, open(os.path.join(GOLD, "patterns.json"), "w"), indent=0)
    print("patterns:", len(cases))


# ----------------------------------------------------------------------------- demo
def gen_demo(ref):
    shutil.copyfile(DEMO_FQ, os.path.join(GOLD, "demo_col0.fastq.gz"))
    shutil.copyfile(DEMO_CSV, os.path.join(GOLD, "demo_telolengths_all.csv"))
    # the log lines that pin the host-side summary (Topsicle_demo/result_justone/topsicle_run.log:21,25-27)
    json.dump(dict(
        pattern="CCCTAAA", slide=6, cutoff=0.7, telophrase=5,
        patterns_line="['AAACC', 'AACCC', 'ACCCT', 'CCCTA', 'CCTAA', 'CTAAA', 'TAAAC', 'TTTGG', 'TTGGG', 'TGGGA', 'GGGAT', 'GGATT', 'GATTT', 'ATTTG']",
        median_line="k-mer: 5, with TRC >= 0.7, median telomere length is 2110.00 bp",
        asymptotic_line="asymptotic TRC, or recommended cutoff: 0.897",
        filtered_line="Median telomere length for reads with TRC cutoff >= 0.897: 2050.00 bp",
    ), open(os.path.join(GOLD, "demo_run_log.json"), "w"), indent=1)

    step1 = []
    for motif, k, minlen, cutoff in [("CCCTAAA", 5, 9000, 0.7), ("CCCTAAA", 5, 9000, -1.0),
                                     ("AAACCCT", 5, 9000, 0.7), ("CCCTAAA", 5, 0, -1.0),
                                     ("CCCTAAA", 4, 0, -1.0), ("CCCTAAA", 7, 0, -1.0),
                                     ("CCCTAA", 4, 9000, 0.3), ("CCCTAAA", 3, 0, -1.0)]:
        rows = ref.patternTRC_count(DEMO_FQ, motif, read_length=minlen, kmer=k, no_bp=1000, cutoff=cutoff)
        step1.append(dict(motif=motif, k=k, min_len=minlen, cutoff=cutoff, rows=rows))
    json.dump(step1, open(os.path.join(GOLD, "demo_step1.json"), "w"))
    print("demo step1 param sets:", len(step1))

    rows = step1[0]["rows"]
    gold = list(csv.reader(open(DEMO_CSV)))[1:]
    assert [g[3] for g in gold] == [r[0] for r in rows]
    pats = ref.patterns_to_search("CCCTAAA", 5)
    out, meta = {}, []
    for i, (rid, pat, tail, trc) in enumerate(rows):
        r = run_step2(ref, DEMO_FQ, rid, pats, 100, 6, 100, 20000, 5, tail)
        assert r["boundary"] == int(gold[i][4]), (rid, r["boundary"], gold[i])
        assert f"{trc:.3f}" == gold[i][2]
        out[f"counts_{i}"] = r["counts"].astype(np.uint8)
        out[f"y_{i}"] = r["y"]
        meta.append(dict(id=rid, tail=tail, trc=trc, best_pattern=pat, boundary=r["boundary"],
                         n_win=int(r["counts"].shape[0])))
    # both tails for two reads (tail=None path), and the default-slide (7) variant of config 1
    for j, i in enumerate([0, 5]):
        rid = rows[i][0]
        for tail in ("forward", "reverse"):
            r = run_step2(ref, DEMO_FQ, rid, pats, 100, 7, 100, 20000, 5, tail)
            out[f"s7_counts_{j}_{tail}"] = r["counts"].astype(np.uint8)
            out[f"s7_y_{j}_{tail}"] = r["y"]
            meta.append(dict(id=rid, tail=tail, slide=7, key=f"s7_{j}_{tail}", boundary=r["boundary"]))
    np.savez_compressed(os.path.join(GOLD, "demo_windows.npz"), **out)
    json.dump(dict(patterns=pats, W=100, slide=6, trimfirst=100, maxlengthtelo=20000, k=5, reads=meta),
              open(os.path.join(GOLD, "demo_windows.json"), "w"), indent=0)
    print("demo step2 reads:", len(rows))


# ----------------------------------------------------------------------------- synthetic
def mutate(rng, seq, sub, ins, dele):
    out = []
    for ch in seq:
        r = rng.random()
        if r < dele:
            continue
        if r < dele + sub:
            out.append("ACGT"[rng.integers(4)])
        else:
            out.append(ch)
        if rng.random() < ins:
            out.append("ACGT"[rng.integers(4)])
    return "".join(out)


def make_read(rng, motif, length, tract, err, reverse):
    phase = int(rng.integers(len(motif)))
    telo = (motif * (tract // len(motif) + 2))[phase:phase + tract]
    rest = "".join("ACGT"[i] for i in rng.integers(0, 4, max(0, length - tract)))
    seq = mutate(rng, telo + rest, *err)
    if reverse:
        seq = seq[::-1].translate(str.maketrans("ACGT", "TGCA"))
    return seq


def gen_synth(ref):
    rng = np.random.default_rng(20250919)
    cases = []

    def add(name, seq, motif, k, W=100, s=6, t=100, M=20000, tail=None, no_bp=1000):
        cases.append(dict(name=name, seq=seq, motif=motif, k=k, W=W, s=s, t=t, M=M, tail=tail, no_bp=no_bp))

    ont, hifi, clean = (0.03, 0.02, 0.02), (0.001, 0.0005, 0.0005), (0, 0, 0)
    # telomeric reads, both orientations, the config patterns
    for i in range(8):
        motif, k, s = [("CCCTAA", 4, 6), ("AAACCCT", 5, 7), ("CCCTAAA", 5, 6), ("CCCTAA", 5, 6)][i % 4]
        add(f"telo{i}", make_read(rng, motif, int(rng.integers(1500, 3200)), int(rng.integers(400, 1400)),
                                  ont if i % 2 else hifi, reverse=bool(i & 2)), motif, k, s=s)
    # multi-k of config 5 incl. self-overlapping k-mers (CTAAC at k=5, CCTAAC/ACCCTA/CTAACC at k=6, CCC at k=3)
    base = make_read(rng, "CCCTAA", 2600, 1200, ont, False)
    for k in (3, 4, 5, 6):
        add(f"multik{k}", base, "CCCTAA", k)
    # self-overlap traps
    trap = ("CTAACTAACTAAC" * 7 + "CCCCCCCCCCCCCCCC" + "ACCCTACCCTACCCTA" * 5 + "CCTAACCTAACTAACCTAAC" * 6 +
            "".join("ACGT"[i] for i in rng.integers(0, 4, 700)))
    for k in (3, 5, 6):
        add(f"trap_k{k}", trap * 2, "CCCTAA", k, t=0, s=1 if k == 5 else 6)
        add(f"trap_rev_k{k}", trap * 2, "CCCTAA", k, t=7, s=5, tail="reverse")
    add("polyC", "C" * 900, "CCCTAA", 3, t=0, s=6)
    add("polyC_k4", "C" * 900, "CCCCCC", 4, t=10, s=7)
    add("AT_rich", ("AT" * 300 + "TA" * 200 + "ATA" * 100), "AT", 2, t=0, s=3, W=20)
    add("homopolymer_motif", "A" * 500 + "ACGT" * 50 + "T" * 300, "A", 1, t=0, s=9, W=30)
    # lower / mixed case, non-ACGT letters
    lc = make_read(rng, "CCCTAA", 1800, 800, ont, False)
    add("lower", lc.lower(), "CCCTAA", 4)
    add("mixed", "".join(c.lower() if rng.random() < 0.5 else c for c in lc), "ccctaa", 4)
    withn = list(make_read(rng, "CCCTAA", 2000, 900, hifi, False))
    for p in rng.integers(0, len(withn), 60):
        withn[p] = "NnRYKMSWU-*."[int(rng.integers(12))]
    add("non_acgt", "".join(withn), "CCCTAA", 4)
    add("non_acgt_rev", "".join(withn), "CCCTAA", 4, tail="reverse", s=7, t=3)
    add("all_N", "N" * 1200, "CCCTAA", 4)
    # short reads / edge lengths (L<1000, L<t+W, L==t+W-1, L==t+W, few windows -> Binseg inadmissible)
    short = make_read(rng, "CCCTAA", 700, 300, clean, False)
    add("short700", short, "CCCTAA", 4)
    for L in (50, 150, 199, 200, 201, 205, 206, 229, 230, 236, 237, 260):
        add(f"edge{L}", short[:L], "CCCTAA", 4)
    add("len1", "C", "CCCTAA", 4)
    add("len_k", "CCCT", "CCCTAA", 4, t=0, W=4, s=1)
    # unusual window / slide / trim / maxlength
    odd = make_read(rng, "AAACCCT", 3000, 1500, ont, True)
    add("odd_W50_s13", odd, "AAACCCT", 5, W=50, s=13, t=33)
    add("odd_W37_s1", odd[:900], "AAACCCT", 5, W=37, s=1, t=0)
    add("odd_W100_s100", odd, "AAACCCT", 5, W=100, s=100, t=100)
    add("odd_W100_s150", odd, "AAACCCT", 5, W=100, s=150, t=100)
    add("odd_M500", odd, "AAACCCT", 5, M=500)
    add("odd_M2000_t250", odd, "AAACCCT", 6, M=2000, t=250, s=4)
    add("odd_k7", odd, "AAACCCT", 7, s=7)
    add("odd_W8_k7", odd[:400], "AAACCCT", 7, W=8, s=2, t=0)
    add("odd_W7_k7", odd[:400], "AAACCCT", 7, W=7, s=2, t=0)      # window text shorter than k
    add("no_bp300", odd, "AAACCCT", 5, no_bp=300)
    add("plant12", make_read(rng, "CTCGGTTATGGG", 2400, 1000, hifi, False), "CTCGGTTATGGG", 8, s=12)
    add("ttagg", make_read(rng, "TTAGG", 2000, 900, ont, True), "TTAGG", 3, s=5)

    tmp = tempfile.mkdtemp()
    arrays, metas = {}, []
    for ci, c in enumerate(cases):
        rid = f"r{ci}"
        path = os.path.join(tmp, f"{rid}.fasta")
        with open(path, "w") as f:
            f.write(f">{rid} synthetic {c['name']}\n{c['seq']}\n")
        pats = ref.patterns_to_search(c["motif"], c["k"])
        # step 1 through the reference, cutoff -1 so every read reports (best pattern, tail, trc)
        rows = ref.patternTRC_count(path, c["motif"], read_length=0, kmer=c["k"], no_bp=c["no_bp"], cutoff=-1.0)
        m = dict(c, id=rid, patterns=pats, step1=rows[0][1:] if rows else None)
        tails = [c["tail"]] if c["tail"] else ([rows[0][2]] if rows else ["forward"])
        if ci % 5 == 0:                                   # every 5th case: pin both tails
            tails = ["forward", "reverse"]
        m["tails"] = tails
        m["boundary"], m["binseg_error"] = {}, {}
        for tail in tails:
            r = run_step2(ref, path, rid, pats, c["W"], c["s"], c["t"], c["M"], c["k"], tail)
            arrays[f"counts_{ci}_{tail}"] = r["counts"].astype(np.uint8)
            arrays[f"starts_{ci}_{tail}"] = r["starts"].astype(np.int32)
            arrays[f"y_{ci}_{tail}"] = r["y"]
            m["boundary"][tail] = r["boundary"]
            m["binseg_error"][tail] = r["binseg_error"]
        metas.append(m)
    shutil.rmtree(tmp)
    np.savez_compressed(os.path.join(GOLD, "synth_cases.npz"), **arrays)
    json.dump(metas, open(os.path.join(GOLD, "synth_cases.json"), "w"))
    print("synthetic cases:", len(cases))


def gen_overview():
    """Rows behind the reference's exploratory plots (descriptive_plot.py:89-165, 233-313) on the demo file:
    the k-mer / following-bases crosstab and the motif positions of the first reads."""
    import re
    import matplotlib.pyplot as plt
    import pandas as pd
    dp = ref_import.load_reference_descriptive_plot()
    out = {"heatmaps": [], "positions": []}
    for motif, k in (("CCCTAAA", 5), ("CCCTAAA", 4), ("AAACCCT", 5)):
        df = dp.patterns_vs_match_heatmap(DEMO_FQ, motif, k, 9000)
        plt.close("all")
        tab = pd.crosstab(df["Match"], df["Pattern"])
        out["heatmaps"].append({"motif": motif, "k": k, "minSeqLength": 9000, "n_rows": int(len(df)),
                                "patterns": [str(c) for c in tab.columns], "matches": [str(i) for i in tab.index],
                                "counts": tab.values.astype(int).tolist(),
                                "first_rows": [[r[0], r[1], list(r[2])] for r in df.head(25).values.tolist()]})
    # descriptive_plot only draws; restate its two searches on the reference's own record reader for 5 reads
    trans = str.maketrans("ACGT", "TGCA")
    for n, (rid, seq) in enumerate(ref_import.read_records(DEMO_FQ)):
        if len(seq) <= 9000:
            continue
        pos = {}
        for pat in ("CCCTAAA", "CCCTAAA".translate(trans)):
            s1, s2 = seq[:9000].upper(), seq[::-1][:9000].upper()
            pos[pat] = [[m.start() for m in re.finditer(re.compile(pat), s1)], [m.start() for m in re.finditer(re.compile(pat), s2)]]
        out["positions"].append({"id": rid, "motif": "CCCTAAA", "minSeqLength": 9000, "pos": pos})
        if len(out["positions"]) == 5:
            break
    json.dump(out, open(os.path.join(GOLD, "demo_overview.json"), "w"))
    print("overview goldens:", [(h["motif"], h["k"], h["n_rows"]) for h in out["heatmaps"]])


CLI_GOLDEN_SEEDS = (104, 108, 126, 132, 134)      # oracle/cli_cases.py: three files / k = 4, 6 / cutoff list; three files / k = 4, 5, 6; --read_check; one file, k = 4 and 6, --rawcountpattern (the raw-count CSVs as digests); three files + raw counts


def gen_cli():
    """Whole runs of the reference's own main() (Topsicle/main.py:312-343 -> analysis_run -> process_file) on seeded inputs:
    the case (input files as text, flags) and what the run left behind (CSV row sequence, summary log lines, filtered files)."""
    import cli_cases
    for seed in CLI_GOLDEN_SEEDS:
        case = cli_cases.make_case(seed)
        with tempfile.TemporaryDirectory() as d:
            inp, out = cli_cases.materialise(case, d)
            code = ref_import.run_reference_main(["-i", inp, "-o", out] + case["argv"])
            assert code == case["exit"], (seed, code)
            expected = cli_cases.normalise(out)
            # the per-read raw-count CSVs (~ 100 kB each) are recorded as digests
            expected["rawcount"] = {f: {"sha256": hashlib.sha256(t.encode()).hexdigest(), "bytes": len(t)} for f, t in expected["rawcount"].items()}
        with open(os.path.join(GOLD, f"cli_{seed}.json"), "w") as h:
            json.dump({"case": case, "expected": expected}, h, indent=0, sort_keys=True)
        print(f"cli_{seed}.json:", 0 if not expected["csv"] else len(expected["csv"]) - 1, "rows,", len(expected["filtered"]), "filtered files,", len(expected["rawcount"]), "raw-count files")


def main():
    os.makedirs(GOLD, exist_ok=True)
    ref = ref_import.load_reference_allsteps()
    gen_patterns(ref)
    gen_demo(ref)
    gen_synth(ref)
    gen_overview()
    gen_cli()
    print("fixture bytes:", sum(os.path.getsize(os.path.join(GOLD, f)) for f in os.listdir(GOLD)))


if __name__ == "__main__":
    main()
