"""Kernel LOGIC checks without a GPU: the HIP kernel source compiled as a sequential host
emulation (tests/emu) must reproduce the reference-generated goldens and the oracle bit for
bit.  The same comparisons run against the real kernels in tests/test_gpu_parity.py."""
import numpy as np
import pytest

import emu_driver as emu
import topsicle_oracle as orc
from topsicle_amd import hiplib, synth

TAILV = {"forward": 0, "reverse": 1}


def check_case(c, arrs, ci, **emu_kw):
    pats = c["patterns"]
    k = c["k"]
    if k > hiplib.MAX_K or len(pats) > hiplib.MAX_PATTERNS:
        return False
    # step 1
    prm = hiplib.make_params(no_bp=c["no_bp"], min_len=0, min_count=-1, window=c["W"], slide=c["s"],
                             trimfirst=c["t"], maxlen=c["M"], flags=hiplib.F_STEP1)
    out = emu.scan(pats, [c["seq"]], prm, **emu_kw)
    cs, ce = orc.trc_counts(c["seq"], pats, c["no_bp"])
    assert out["c_start"][0].tolist() == cs, c["name"]
    assert out["c_end"][0].tolist() == ce, c["name"]
    r = out["results"][0]
    if c["step1"] is not None:
        pat, tail, trc = c["step1"]
        assert TAILV[tail] == r["tail"], c["name"]
        best, idx = (r["best_start"], r["best_start_idx"]) if r["tail"] == 0 else (r["best_end"], r["best_end_idx"])
        assert pats[idx] == pat and best / (c["no_bp"] / len(c["motif"])) == trc, c["name"]
    # step 2 + 3 for every pinned tail
    for tail in c["tails"]:
        want = arrs[f"counts_{ci}_{tail}"].astype(np.int64)
        prm = hiplib.make_params(window=c["W"], slide=c["s"], trimfirst=c["t"], maxlen=c["M"],
                                 flags=hiplib.F_WINDOWS | hiplib.F_BINSEG | hiplib.F_TAILS_IN | hiplib.F_STORE_SUMS | hiplib.F_STORE_RAW)
        out = emu.scan(pats, [c["seq"]], prm, tails=[TAILV[tail]], **emu_kw)
        assert out["raw"].shape[0] == want.shape[0], c["name"]
        if want.shape[0]:
            assert np.array_equal(out["raw"], want), (c["name"], tail)
            assert np.array_equal(out["sums"], want.sum(axis=1)), (c["name"], tail)
        # sums-only fast path must agree with the exact path
        prm.flags = hiplib.F_WINDOWS | hiplib.F_BINSEG | hiplib.F_TAILS_IN | hiplib.F_STORE_SUMS
        out2 = emu.scan(pats, [c["seq"]], prm, tails=[TAILV[tail]], **emu_kw)
        assert np.array_equal(out2["sums"], want.sum(axis=1) if want.shape[0] else np.zeros(0)), (c["name"], tail)
        r = out2["results"][0]
        assert r["n_win"] == want.shape[0]
        b = c["boundary"][tail]
        if b is None:
            assert r["bkp"] == -1, c["name"]
        else:
            exact = orc.binseg_l2_exact(want.sum(axis=1))
            assert r["bkp"] == exact, c["name"]       # the kernel's own answer: the exact rule (ties -> the larger index)
            if r["bkp"] * c["s"] + c["t"] != b:      # only legal where float64 cannot resolve a tie -- and then the read is flagged
                assert c["name"] in ("polyC",) and (r["flags"] & hiplib.RES_TIE), c["name"]
            if r["flags"] & hiplib.RES_TIE:          # ... and ruptures' float64 arithmetic on the read's S_w gives the reference's boundary
                assert hiplib.binseg_l2_float64(out2["sums"] / len(pats)) * c["s"] + c["t"] == b, c["name"]
    return True


@pytest.mark.parametrize("force_generic", [0, 1])
def test_emulation_synthetic_goldens(synth_cases, force_generic):
    meta, arrs = synth_cases
    done = sum(check_case(c, arrs, ci, force_generic=force_generic) for ci, c in enumerate(meta))
    assert done >= 50


def test_plan_picks_specialised_kernels_for_the_baseline_configs():
    for k, P, slide, nwin, variant in [(4, 12, 6, 2467, 6), (5, 14, 7, 2829, 7), (4, 12, 6, 3301, 6), (5, 12, 6, 3301, 6),
                                       (6, 12, 6, 3301, 6), (3, 10, 5, 3000, 5), (6, 16, 8, 2400, 8), (6, 17, 8, 2400, 0), (4, 12, 11, 1000, 11), (4, 12, 13, 1000, 0), (4, 12, 3, 1000, 3), (4, 12, 2, 1000, 0)]:
        pl = emu.plan(k, P, hiplib.make_params(slide=slide), nwin)
        assert pl["variant"] == variant, (k, P, slide, pl)
        assert pl["lds_bytes"] <= 160 * 1024
    assert emu.plan(4, 12, hiplib.make_params(slide=6), 2467, force_generic=1)["variant"] == 0


def test_emulation_small_tiles_and_misalignment(synth_cases):
    """Force many tiles per read (2..8 spans) and every 16-byte misalignment class."""
    meta, arrs = synth_cases
    rng = np.random.default_rng(3)
    for ci, c in enumerate(meta):
        if ci % 3 == 0:
            check_case(c, arrs, ci, spans_pref=int(rng.integers(1, 9)), base_shift=int(rng.integers(0, 16)))


def test_emulation_demo_reads(demo_windows, demo_records):
    meta, arrs = demo_windows
    seqs = dict(demo_records)
    pats = meta["patterns"]
    reads = [r for r in meta["reads"] if "key" not in r]
    prm = hiplib.make_params(min_len=9000, min_count=int(0.7 * (1000 / 7)), window=100, slide=6, trimfirst=100,
                             maxlen=20000, flags=hiplib.F_STEP1 | hiplib.F_WINDOWS | hiplib.F_BINSEG | hiplib.F_STORE_SUMS)
    out = emu.scan(pats, [seqs[r["id"]] for r in reads], prm)
    for i, r in enumerate(reads):
        res = out["results"][i]
        assert res["pass"] == 1 and res["tail"] == TAILV[r["tail"]]
        best = res["best_start"] if res["tail"] == 0 else res["best_end"]
        assert best / (1000 / 7) == r["trc"]
        lo, hi = out["win_off"][i], out["win_off"][i + 1]
        assert np.array_equal(out["sums"][lo:hi], arrs[f"counts_{i}"].astype(np.int64).sum(axis=1))
        assert res["bkp"] * 6 + 100 == r["boundary"]


def test_emulation_whole_demo_file_filter(demo_records, gold_dir):
    """All 44 demo reads in one batch: exactly the 17 golden reads pass the filter."""
    import csv, os
    pats = orc.kmer_table("CCCTAAA", 5)
    ratio = 1000 / 7
    min_count = max(c for c in range(0, 1001) if not (c / ratio > 0.7))
    prm = hiplib.make_params(min_len=9000, min_count=min_count, flags=hiplib.F_STEP1 | hiplib.F_WINDOWS | hiplib.F_BINSEG)
    out = emu.scan(pats, [s for _, s in demo_records], prm, base_shift=5)
    gold = list(csv.reader(open(os.path.join(gold_dir, "demo_telolengths_all.csv"))))[1:]
    got = [(demo_records[i][0], int(r["bkp"]) * 6 + 100) for i, r in enumerate(out["results"]) if r["pass"]]
    assert got == [(g[3], int(g[4])) for g in gold]


@pytest.mark.parametrize("seed", range(18))
def test_emulation_random_vs_oracle(seed):
    """Seeded random reads / parameters against the Python oracle (counts bit-exact).  Covers the
    specialised slides (5..8) with and without self-overlapping k-mers and invalid bases, and
    the generic path."""
    rng = np.random.default_rng(100 + seed)
    motif, k = [("CCCTAA", 4), ("CCCTAA", 5), ("AAACCCT", 5), ("CCCTAA", 6), ("TTAGGG", 3), ("AAACCCT", 7)][seed % 6]
    pats = orc.kmer_table(motif, k)
    W = int(rng.choice([100, 64, 23, 100]))
    s = [6, 7, 5, 8, 6, 7, 1, 4, 16, 11, 8, 5, 6, 7, 5, 8, 3, 12][seed]
    if s in (5, 6, 7, 8) and seed < 14:
        W = int(rng.choice([100, 100, 120, 90]))        # wide enough for the fused kernels (q >= 8)
    t = int(rng.choice([100, 0, 17]))
    M = int(rng.choice([20000, 900, 1500]))
    seqs, tails = [], []
    for i in range(6):
        L = int(rng.integers(0, 2600))
        tract = int(rng.integers(0, max(1, L // 2)))
        ph = int(rng.integers(len(motif)))
        body = (motif * (tract // len(motif) + 2))[ph:ph + tract] + "".join("ACGT"[x] for x in rng.integers(0, 4, max(0, L - tract)))
        body = list(body[:L])
        for p in rng.integers(0, max(1, len(body)), len(body) // 25):      # errors, N, lower case
            if len(body):
                body[p] = "ACGTNacgtn"[int(rng.integers(10))]
        seq = "".join(body)
        if rng.random() < 0.5:
            seq = seq[::-1].translate(str.maketrans("ACGTacgt", "TGCAtgca"))
        seqs.append(seq)
        tails.append(int(rng.integers(2)))
    prm = hiplib.make_params(window=W, slide=s, trimfirst=t, maxlen=M,
                             flags=hiplib.F_WINDOWS | hiplib.F_BINSEG | hiplib.F_TAILS_IN | hiplib.F_STORE_SUMS | hiplib.F_STORE_RAW)
    spans_pref = 0 if seed % 3 else int(rng.integers(1, 6))      # 0 = planner's choice (fused where possible)
    out = emu.scan(pats, seqs, prm, tails=tails, spans_pref=spans_pref, base_shift=int(rng.integers(16)))
    prm1 = hiplib.make_params(no_bp=1000, flags=hiplib.F_STEP1)
    out1 = emu.scan(pats, seqs, prm1)
    for i, seq in enumerate(seqs):
        cs, ce = orc.trc_counts(seq, pats)
        assert out1["c_start"][i].tolist() == cs and out1["c_end"][i].tolist() == ce
        _, counts = orc.window_count_matrix(seq, ["forward", "reverse"][tails[i]], pats, W, s, t, M)
        lo, hi = out["win_off"][i], out["win_off"][i + 1]
        assert hi - lo == counts.shape[0]
        assert np.array_equal(out["raw"][lo:hi], counts.reshape(-1, len(pats)))
        assert np.array_equal(out["sums"][lo:hi], counts.sum(axis=1))
        want = orc.binseg_l2_exact(counts.sum(axis=1)) if counts.shape[0] else None
        assert out["results"][i]["bkp"] == (-1 if want is None else want)


@pytest.mark.parametrize("case", [(6, 100, "CCCTAA", 4), (7, 100, "AAACCCT", 5), (5, 200, "CCCTAA", 4), (8, 260, "CCCTAA", 4)])
def test_emulation_multi_tile_fused(case):
    """Reads long enough for several fused tiles: the carried candidate sums (16-bit tile-relative for the
    default window, 32-bit for wide windows) and the tile seams."""
    s, W, motif, k = case
    rng = np.random.default_rng(7 * s + W)
    pats = orc.kmer_table(motif, k)
    seqs, tails = [], []
    for i in range(3):
        L = int(rng.integers(5000, 9000))
        tract = int(rng.integers(500, 4000))
        body = list((motif * (tract // len(motif) + 2))[:tract] + "".join("ACGT"[x] for x in rng.integers(0, 4, L - tract)))
        for p in rng.integers(0, L, L // 12):
            body[p] = "ACGT"[int(rng.integers(4))]
        if i == 2:
            body[int(rng.integers(L))] = "N"
        seqs.append("".join(body))
        tails.append(0)
    prm = hiplib.make_params(window=W, slide=s, trimfirst=100, maxlen=20000,
                             flags=hiplib.F_WINDOWS | hiplib.F_BINSEG | hiplib.F_TAILS_IN | hiplib.F_STORE_SUMS)
    out = emu.scan(pats, seqs, prm, tails=tails)
    for i, seq in enumerate(seqs):
        _, counts = orc.window_count_matrix(seq, "forward", pats, W, s, 100, 20000)
        lo, hi = out["win_off"][i], out["win_off"][i + 1]
        assert hi - lo == counts.shape[0] and hi - lo > 500
        assert np.array_equal(out["sums"][lo:hi], counts.sum(axis=1))
        assert out["results"][i]["bkp"] == orc.binseg_l2_exact(counts.sum(axis=1))


@pytest.mark.parametrize("traps", [3, 400])
def test_emulation_self_overlap_recount_paths(traps):
    """Tables with self-overlapping k-mers (CCCTAA at k=5: CTAAC / GATTG have period 4), sums only: a few
    flagged windows per tile take the whole-wave recount, many take the per-lane recount."""
    rng = np.random.default_rng(traps)
    pats = orc.kmer_table("CCCTAA", 5)
    seqs = []
    for i in range(2):
        L = 7000
        body = list(("CCCTAA" * 700)[:3000] + "".join("ACGT"[x] for x in rng.integers(0, 4, L - 3000)))
        for p in rng.integers(0, L - 20, traps):
            body[p:p + 13] = list("CTAACTAACTAAC" if rng.random() < 0.5 else "GATTGATTGATTG")
        if i:
            body[int(rng.integers(L))] = "N"
        else:
            body[200:213] = list("CTAACTAACTAAC")            # overlapping occurrences inside the step-1 head
            body[L - 300:L - 287] = list("GTTAGTTAGTTAG")    # ... and inside the reversed end head (GATTG reversed)
        seqs.append("".join(body[:L]))
    # step 1: packed counters + conflict mask + greedy recount of the conflicting patterns (read 0, no N),
    # histogram path (read 1, holds an N)
    out1 = emu.scan(pats, seqs, hiplib.make_params(no_bp=1000, flags=hiplib.F_STEP1))
    for i, seq in enumerate(seqs):
        cs, ce = orc.trc_counts(seq, pats)
        assert out1["c_start"][i].tolist() == cs and out1["c_end"][i].tolist() == ce
    prm = hiplib.make_params(window=100, slide=6, trimfirst=100, maxlen=20000,
                             flags=hiplib.F_WINDOWS | hiplib.F_BINSEG | hiplib.F_TAILS_IN | hiplib.F_STORE_SUMS)
    out = emu.scan(pats, seqs, prm, tails=[0, 1])
    for i, seq in enumerate(seqs):
        _, counts = orc.window_count_matrix(seq, ["forward", "reverse"][i], pats, 100, 6, 100, 20000)
        lo, hi = out["win_off"][i], out["win_off"][i + 1]
        assert np.array_equal(out["sums"][lo:hi], counts.sum(axis=1))
        assert out["results"][i]["bkp"] == orc.binseg_l2_exact(counts.sum(axis=1))


@pytest.mark.parametrize("slide", [5, 6, 8])
def test_emulation_long_period_table(slide):
    """CCCTAA at k=7: CCCTAAC / TAACCCT ... have period 6.  Slide 5 cannot see p + 6 from the next block, so
    the planner must fall back to the generic kernel there; slides 6 and 8 keep the fused kernel."""
    rng = np.random.default_rng(slide)
    pats = orc.kmer_table("CCCTAA", 7)
    L = emu.lib()
    seqs = []
    for i in range(2):
        body = list(("CCCTAA" * 500)[:2400] + "".join("ACGT"[x] for x in rng.integers(0, 4, 2600)))
        for p in rng.integers(0, 2300, 12):
            del body[p]                                   # deletions inside the repeat create overlapping 7-mers
        seqs.append("".join(body))
    prm = hiplib.make_params(window=100, slide=slide, trimfirst=100, maxlen=20000,
                             flags=hiplib.F_WINDOWS | hiplib.F_BINSEG | hiplib.F_TAILS_IN | hiplib.F_STORE_SUMS)
    before = {v: L.emu_variant_calls(v) for v in (0, 5, 6, 8)}
    out = emu.scan(pats, seqs, prm, tails=[0, 0])
    used = [v for v in (0, 5, 6, 8) if L.emu_variant_calls(v) > before[v]]
    assert used == ([0] if slide == 5 else [slide])
    for i, seq in enumerate(seqs):
        _, counts = orc.window_count_matrix(seq, "forward", pats, 100, slide, 100, 20000)
        lo, hi = out["win_off"][i], out["win_off"][i + 1]
        assert np.array_equal(out["sums"][lo:hi], counts.sum(axis=1))
    out1 = emu.scan(pats, seqs, hiplib.make_params(no_bp=1000, flags=hiplib.F_STEP1))
    for i, seq in enumerate(seqs):
        cs, ce = orc.trc_counts(seq, pats)
        assert out1["c_start"][i].tolist() == cs and out1["c_end"][i].tolist() == ce


def test_emulation_binseg_standalone():
    rng = np.random.default_rng(5)
    sums, off = [], [0]
    for n in [0, 3, 6, 7, 8, 12, 50, 255, 256, 257, 1000, 3301]:
        v = rng.integers(12, 200, n)
        if n > 20:
            v[: n // 3] += 150
        sums.append(v)
        off.append(off[-1] + n)
    allv = np.concatenate(sums).astype(np.int32)
    bkp, gain = emu.binseg(allv, np.array(off, np.int64), 12)
    for i, v in enumerate(sums):
        want = orc.binseg_l2_exact(v)
        assert bkp[i] == (-1 if want is None else want)
        if want is not None:
            _, g = orc.binseg_l2_numpy(v / 12)
            bf, _ = orc.binseg_l2_numpy(v / 12)
            if bf == want:
                assert abs(gain[i] - g) <= 1e-9 * max(1.0, abs(g))


def test_zz_specialised_paths_were_exercised():
    """Bookkeeping: the runs above went through every specialised instantiation and the generic one."""
    L = emu.lib()
    calls = {v: L.emu_variant_calls(v) for v in (0, 5, 6, 7, 8)}
    assert all(c > 0 for c in calls.values()), calls


@pytest.mark.parametrize("motif,k,slide", [("CCCTAACCTA", 8, 10), ("TTAGGGTTAGGCA", 11, 6), ("AAAACCCCTT", 9, 7)])
def test_emulation_long_kmers_hashed_table(motif, k, slide):
    """k > 7: the 4^k table does not fit LDS, the generic kernel uses a perfect hash of the pattern codes.
    (AAAACCCCTT at k=9 doubles to k-mers with long self-overlap periods.)"""
    rng = np.random.default_rng(k)
    pats = orc.kmer_table(motif, k)
    seqs = []
    for i in range(3):
        L = int(rng.integers(1500, 3500))
        tract = int(rng.integers(300, 1400))
        body = list((motif * (tract // len(motif) + 2))[:tract] + "".join("ACGT"[x] for x in rng.integers(0, 4, L - tract)))
        for p in rng.integers(0, L, L // 30):
            body[p] = "ACGTNacgt"[int(rng.integers(9))]
        seqs.append("".join(body))
    prm = hiplib.make_params(window=100, slide=slide, trimfirst=100, maxlen=20000,
                             flags=hiplib.F_WINDOWS | hiplib.F_BINSEG | hiplib.F_TAILS_IN | hiplib.F_STORE_SUMS | hiplib.F_STORE_RAW)
    out = emu.scan(pats, seqs, prm, tails=[0, 1, 0])
    out1 = emu.scan(pats, seqs, hiplib.make_params(no_bp=1000, flags=hiplib.F_STEP1))
    for i, seq in enumerate(seqs):
        cs, ce = orc.trc_counts(seq, pats)
        assert out1["c_start"][i].tolist() == cs and out1["c_end"][i].tolist() == ce
        _, counts = orc.window_count_matrix(seq, ["forward", "reverse"][i & 1], pats, 100, slide, 100, 20000)
        lo, hi = out["win_off"][i], out["win_off"][i + 1]
        assert hi - lo == counts.shape[0]
        assert np.array_equal(out["raw"][lo:hi], counts.reshape(-1, len(pats)))
        assert np.array_equal(out["sums"][lo:hi], counts.sum(axis=1))


@pytest.mark.parametrize("jump,min_size,slide", [(1, 2, 6), (3, 2, 6), (8, 4, 7), (1, 1, 11), (13, 2, 6)])
def test_emulation_other_jump_values(jump, min_size, slide):
    """The change-point candidates b = c * jump for jumps other than ruptures' default 5 (jump = 1 has no 32-bit
    reciprocal: found by the GPU sweep tests/gpu_fuzz.py)."""
    rng = np.random.default_rng(jump * 31 + slide)
    pats = orc.kmer_table("CCCTAA", 4)
    seqs = []
    for i in range(3):
        L = int(rng.integers(1500, 7000))
        tract = int(rng.integers(200, 1400))
        body = list(("CCCTAA" * 300)[:tract] + "".join("ACGT"[x] for x in rng.integers(0, 4, L - tract)))
        for p in rng.integers(0, L, L // 20):
            body[p] = "ACGT"[int(rng.integers(4))]
        seqs.append("".join(body))
    prm = hiplib.make_params(window=100, slide=slide, trimfirst=100, maxlen=20000, jump=jump, min_size=min_size,
                             flags=hiplib.F_WINDOWS | hiplib.F_BINSEG | hiplib.F_TAILS_IN | hiplib.F_STORE_SUMS)
    out = emu.scan(pats, seqs, prm, tails=[0, 0, 0])
    for i, seq in enumerate(seqs):
        _, counts = orc.window_count_matrix(seq, "forward", pats, 100, slide, 100, 20000)
        lo, hi = out["win_off"][i], out["win_off"][i + 1]
        assert np.array_equal(out["sums"][lo:hi], counts.sum(axis=1))
        want = orc.binseg_l2_exact(counts.sum(axis=1), jump, min_size)
        assert out["results"][i]["bkp"] == (-1 if want is None else want), (jump, min_size, i)


def _pp_reads(rng, motif, k, n, L, chain_units):
    """Telomere-like reads peppered with single-base deletions (pairs of overlapping k-mers), short and long runs
    of the k-mers' own period (chains of 3+ overlapping occurrences) and random sequence."""
    pats = orc.kmer_table(motif, k)
    seqs = []
    for i in range(n):
        body = []
        while len(body) < L:
            u = rng.random()
            if u < 0.55:
                rep = list(motif * int(rng.integers(3, 60)))
                for _ in range(int(rng.integers(0, 6))):          # deletions / substitutions inside the repeat
                    p = int(rng.integers(0, len(rep)))
                    if rng.random() < 0.7:
                        del rep[p]
                    else:
                        rep[p] = "ACGT"[int(rng.integers(4))]
                body += rep
            elif u < 0.75 and chain_units:
                unit = chain_units[int(rng.integers(len(chain_units)))]
                body += list(unit * int(rng.integers(2, 40)))       # (CCTAA)n-like run: one long chain
            else:
                body += ["ACGT"[x] for x in rng.integers(0, 4, int(rng.integers(5, 400)))]
        seqs.append("".join(body[:L]))
    return pats, seqs


@pytest.mark.parametrize("motif,k,slide,units", [
    ("CCCTAA", 5, 6, ["CTAA", "GATT"]),          # CTAAC / GATTG: period 4
    ("CCCTAA", 6, 6, ["CCTAA", "GGATT", "CTAAC"]),   # CCTAAC, CTAACC + complements: period 5
    ("CCCTAA", 6, 5, ["CCTAA"]),
    ("CCCTAA", 6, 7, ["CCTAA", "GGATT"]),
    ("CCCTAA", 6, 8, ["CCTAA"]),
    ("TTTAGGG", 7, 6, ["TTTAGG", "AAATCC"]),     # period 6 (GGTTTAG ...)
    ("TTAGGG", 5, 5, ["TTAG", "AATC"]),           # GTTAG: period 4
    ("CCCTAA", 4, 6, []),                         # no self-overlap: raw counts only
    ("AAACCCT", 5, 7, []),
])
def test_emulation_per_pattern_tiles(motif, k, slide, units):
    """tile_pp_s (raw rows) and tile_so_s (sums only): exact counts without recounting (canonical picks + start skips),
    both tails, several tiles per read, reads with and without non-ACGT letters."""
    rng = np.random.default_rng(sum(map(ord, motif)) * 1000 + 10 * k + slide)
    pats, seqs = _pp_reads(rng, motif, k, 4, 9000, units)
    seqs[3] = seqs[3][:4000] + "N" + seqs[3][4001:]                  # one tile of read 3 falls back to the recount path
    so = any(p[:d] == p[-d:] for p in pats for d in range(1, k))
    L = emu.lib()
    tails = [0, 1, 0, 1]
    for raw in (1, 0):
        if not raw and not so:
            continue
        flags = hiplib.F_WINDOWS | hiplib.F_BINSEG | hiplib.F_TAILS_IN | hiplib.F_STORE_SUMS | (hiplib.F_STORE_RAW if raw else 0)
        prm = hiplib.make_params(window=100, slide=slide, trimfirst=100, maxlen=20000, flags=flags)
        # raw rows: per-pattern tiles (tile_pp_s, counter 0); sums only, self-overlap table: tile_so_s (5) for tiles with a
        # chained occurrence, the chain-detecting plain tile (6) for the others
        count = (lambda: L.emu_counter(0)) if raw else (lambda: L.emu_counter(5) + L.emu_counter(6))
        t0, r0, f0 = count(), L.emu_counter(1), L.emu_counter(6)
        out = emu.scan(pats, seqs, prm, tails=tails, base_shift=int(rng.integers(16)))
        tiles, redone, fast = count() - t0, L.emu_counter(1) - r0, L.emu_counter(6) - f0
        assert tiles >= 8, "the canonical-pick tiles were not used"
        if not raw:
            assert 0 < fast < tiles, ("both the chain-free and the chained tile path should have run", fast, tiles)
        nwin_total = 0
        for i, seq in enumerate(seqs):
            _, counts = orc.window_count_matrix(seq, ["forward", "reverse"][tails[i]], pats, 100, slide, 100, 20000)
            lo, hi = out["win_off"][i], out["win_off"][i + 1]
            nwin_total += hi - lo
            assert hi - lo == counts.shape[0]
            assert np.array_equal(out["sums"][lo:hi], counts.sum(axis=1)), (i, raw)
            if raw:
                assert np.array_equal(out["raw"][lo:hi], counts), (i, raw)
            assert out["results"][i]["bkp"] == orc.binseg_l2_exact(counts.sum(axis=1))
        # recounts only where three or more occurrences chain (the planted runs): far fewer than the windows
        assert redone < nwin_total // 2, (redone, nwin_total)
        if not units:
            assert redone == 0


@pytest.mark.parametrize("motif,k,slide,unit", [
    ("CCCTAA", 5, 6, "CTAA"),                     # CTAAC: period 4
    ("CCCTAA", 6, 6, "CCTAA"),                    # CCTAAC: period 5
    ("TTTAGGG", 7, 7, "TTTAGG"),                  # period 6
])
@pytest.mark.parametrize("detect_every_tile", [True, False])
def test_emulation_chain_free_tiles_hand_over(motif, k, slide, unit, detect_every_tile, monkeypatch):
    """Self-overlap table: tiles without a chained occurrence complete as plain tiles (sums only: tile_fused_s<.., CD>; raw rows:
    tile_pp_s<S, 0, CD>), the others go through tile_so_s / tile_pp_s<S, D>.  Chains planted around every tile boundary (before, across, after; 2 to 6 links; both
    tails) check the hand-over in both directions: what a plain tile leaves for a chained successor, and a chained tile
    followed by a plain one.  detect_every_tile (planner knob so_order = 2): the raw-row kernels try the chain-free tile first for EVERY tile (the
    order of round 3); otherwise a tile that follows a chained one goes straight to the canonical-pick tile (round 4)."""
    if detect_every_tile:
        monkeypatch.setitem(emu.KNOBS, "so_order", 2)
    rng = np.random.default_rng(7 * k + slide)
    pats = orc.kmer_table(motif, k)
    flags = hiplib.F_WINDOWS | hiplib.F_BINSEG | hiplib.F_TAILS_IN | hiplib.F_STORE_SUMS
    prm = hiplib.make_params(window=100, slide=slide, trimfirst=100, maxlen=20000, flags=flags)
    tw = emu.plan_table(pats, prm, 3301)["tw"]
    L = 100 + 2 * tw * slide + 1500
    d = len(unit)
    seqs, tails = [], []
    for off in range(-3 * d - 2, 3 * d + 3):
        body = ["ACGT"[x] for x in rng.integers(0, 4, L)]
        for t in (1, 2):
            run = list(unit * int(rng.integers(2, 7)) + unit[:k - d])
            at = 100 + t * tw * slide + off
            body[at:at + len(run)] = run
        tail = int(rng.integers(2))
        seqs.append("".join(body if tail == 0 else body[::-1]))
        tails.append(tail)
    lib = emu.lib()
    for raw in (0, 1):
        # sums only: chain-free tiles (counter 6) vs tile_so_s (5); raw rows: chain-free per-pattern tiles (7) vs all of them (0)
        p2 = hiplib.make_params(window=100, slide=slide, trimfirst=100, maxlen=20000, flags=flags | (hiplib.F_STORE_RAW if raw else 0))
        fi, si = (7, 0) if raw else (6, 5)
        f0, s0, r0 = lib.emu_counter(fi), lib.emu_counter(si), lib.emu_counter(1)
        out = emu.scan(pats, seqs, p2, tails=tails, base_shift=int(rng.integers(16)))
        fast, slow, redone = lib.emu_counter(fi) - f0, lib.emu_counter(si) - s0, lib.emu_counter(1) - r0
        if raw:
            slow -= fast                                # (counter 0 counts every completed per-pattern tile)
        assert slow > 0 and (fast > 0 or (raw and not detect_every_tile)), (raw, fast, slow)
        for i, seq in enumerate(seqs):
            _, counts = orc.window_count_matrix(seq, ["forward", "reverse"][tails[i]], pats, 100, slide, 100, 20000)
            lo, hi = out["win_off"][i], out["win_off"][i + 1]
            assert hi - lo == counts.shape[0]
            bad = np.flatnonzero(out["sums"][lo:hi] != counts.sum(axis=1))
            assert bad.size == 0, (raw, i, tails[i], bad[:8], tw)
            if raw:
                assert np.array_equal(out["raw"][lo:hi], counts.reshape(-1, len(pats))), (i, tails[i])
        assert redone < 40 * len(seqs)


@pytest.mark.parametrize("motif,k,slide,counts", [("CCCTAA", 5, 6, [1, 2, 3, 5, 6, 7, 9, 10, 11, 12]), ("AAACCCT", 5, 7, [13, 14])])
def test_emulation_raw_rows_for_any_number_of_patterns(motif, k, slide, counts):
    """Raw rows of the per-pattern tiles leave through LDS whatever the row length: 4 / 8 / 12 bytes packed, other even
    lengths in 16-bit units, odd lengths (hand-made pattern lists through the C ABI) byte by byte."""
    rng = np.random.default_rng(k * 100 + slide)
    table = orc.kmer_table(motif, k)
    _, seqs = _pp_reads(rng, motif, k, 3, 7000, [])
    flags = hiplib.F_WINDOWS | hiplib.F_TAILS_IN | hiplib.F_STORE_SUMS | hiplib.F_STORE_RAW
    prm = hiplib.make_params(window=100, slide=slide, trimfirst=100, maxlen=20000, flags=flags)
    L = emu.lib()
    tails = [0, 1, 0]
    for P in counts:
        pats = table[:P]
        t0 = L.emu_counter(0)
        out = emu.scan(pats, seqs, prm, tails=tails)
        assert L.emu_counter(0) > t0, ("per-pattern tiles not used", P)
        for i, seq in enumerate(seqs):
            _, want = orc.window_count_matrix(seq, ["forward", "reverse"][tails[i]], pats, 100, slide, 100, 20000)
            lo, hi = out["win_off"][i], out["win_off"][i + 1]
            assert np.array_equal(out["raw"][lo:hi], want.reshape(-1, P)), (P, i)
            assert np.array_equal(out["sums"][lo:hi], want.sum(axis=1)), (P, i)


def test_planner_picks_per_pattern_tiles():
    """Which tables take the per-pattern tiles: one self-overlap period (or none), distinct k-mers, k >= 4."""
    def plan(motif, k, slide, flags=0):
        return emu.plan_table(orc.kmer_table(motif, k), hiplib.make_params(slide=slide, flags=hiplib.F_WINDOWS | flags), 3301)
    assert plan("CCCTAA", 4, 6)["pp_d"] == 0 and plan("CCCTAA", 4, 6)["pair_n"] == 1024          # no overlap; pair table for sums only
    assert plan("CCCTAA", 4, 6, hiplib.F_STORE_RAW)["pair_n"] == 1024                            # raw counts (round 5): a pair table of fields at k = 4, slide 6
    assert plan("CCCTAA", 4, 7, hiplib.F_STORE_RAW)["pair_n"] == 0 and plan("CCCTAA", 4, 5, hiplib.F_STORE_RAW)["pair_n"] == 0    # ... only
    assert plan("CCCTAA", 5, 6)["pp_d"] == 4 and plan("CCCTAA", 6, 6)["pp_d"] == 5               # CTAAC: period 4; CCTAAC: period 5
    assert plan("TTTAGGG", 7, 7)["pp_d"] == 6 and plan("AAACCCT", 5, 7)["pp_d"] == 0
    assert plan("ACACAC", 5, 6)["pp_d"] == -1                                                    # ACACA: periods 2 and 4
    assert plan("CCCTAA", 3, 6)["pp_d"] == -1                                                    # k < 4: too many occurrences per lane
    assert plan("CCCTAA", 6, 4)["variant"] == 0                                                  # no fused kernel for this slide
    for motif, k, s in [("CCCTAA", 4, 6), ("CCCTAA", 5, 6), ("AAACCCT", 5, 7)]:
        assert plan(motif, k, s)["lds_bytes"] <= 32000                                           # five workgroups per CU


def test_dispatch_order_longest_reads_first_and_file_order_for_equal_reads():
    """tps::plan_dispatch_order (round 5): the read each wave slot of a launch takes -- classes of equal work, the longest first, reads the
    length filter drops last, file order inside a class; a batch of one class keeps file order (no indirection in the kernel)."""
    rng = np.random.default_rng(5)
    assert len(emu.dispatch_order(np.full(1000, 2484), np.ones(1000))) == 0                       # config 2: equal reads
    assert len(emu.dispatch_order(np.zeros(50), np.zeros(50))) == 0 and len(emu.dispatch_order([7], [1])) == 0
    assert len(emu.dispatch_order(2484 - rng.integers(0, 20, 1000), np.ones(1000))) == 0          # lengths within one class of 64
    lens = np.clip(np.exp(rng.normal(9.3, 0.8, 5000)), 60, 60000).astype(np.int64)                # an ONT file's lengths
    n_win = np.array([hiplib.window_count(int(x), 100, 6, 100, 20000) for x in lens])
    passes = lens > 9000
    order = emu.dispatch_order(n_win, passes)
    assert sorted(order.tolist()) == list(range(5000))                                            # a permutation
    work = np.where(passes, n_win, -1)[order]
    mx = n_win[passes].max()
    cls = np.where(work < 0, 0, 1 + (work * 62 + mx - 1) // mx)
    assert np.all(np.diff(cls) <= 0)                                                              # classes in descending order
    for c in np.unique(cls):
        assert np.all(np.diff(order[cls == c]) > 0)                                               # file order inside a class
    assert not passes[order[-(~passes).sum():]].any() and passes[order[:passes.sum()]].all()      # the dropped reads leave last


def test_emulation_results_do_not_depend_on_the_dispatch_order():
    """The emulation runs the reads in the library's dispatch order: a ragged batch (reordered) against the same reads one by one."""
    motif, k = "CCCTAA", 4
    pats = orc.kmer_table(motif, k)
    b, o, _ = synth.make_ragged_reads(40, motif, 77, len_mu=8.6, len_sigma=0.9, max_len=14000)
    seqs = synth.split_reads(b, o)
    prm = hiplib.make_params(min_len=3000, min_count=20, slide=6, flags=hiplib.F_STEP1 | hiplib.F_WINDOWS | hiplib.F_BINSEG | hiplib.F_STORE_SUMS)
    lens = np.diff(o)
    order = emu.dispatch_order([hiplib.window_count(int(x), 100, 6, 100, 20000) for x in lens], lens > 3000)
    assert len(order) == 40 and not np.array_equal(order, np.arange(40))
    out = emu.scan(pats, seqs, prm)
    for i in (int(order[0]), int(order[-1]), 7, 23):
        one = emu.scan(pats, [seqs[i]], prm)
        for f in ("pass", "tail", "n_win", "bkp", "best_start", "best_end"):
            assert out["results"][f][i] == one["results"][f][0], (i, f)
        assert np.array_equal(out["sums"][out["win_off"][i]:out["win_off"][i + 1]], one["sums"])


@pytest.mark.parametrize("motif,k,slide", [("CCCTAA", 4, 6), ("CCCTAA", 4, 5), ("TTAGGG", 4, 7), ("TTAGGC", 4, 6), ("CCCTAA", 5, 6), ("CCCTAA", 6, 6), ("CCCTAA", 6, 5),
                                           ("TTAGGG", 5, 7)])
def test_emulation_per_pattern_tiles_store_every_second_row(motif, k, slide):
    """Strided scans at twice a base slide (round 5): the per-pattern tiles store the even windows' raw rows themselves, packed per read in
    the layout of slide 2 s (ScanArgs::raw_m = 2) -- tables without self-overlap and, with the repairs of chained windows mapped the same
    way, with one (k = 5, 6 of a 6-mer motif: ONT-like errors make chains) -- against the oracle AT slide 2 s, reads of one, several and
    partly filled tiles; the sums stay those of the base slide."""
    pats = orc.kmer_table(motif, k)
    P = len(pats)
    if P % 4:
        pytest.skip("rows of whole dwords only")
    rng = np.random.default_rng(slide)
    seqs = []
    for L in (1500, 3300, 7000, 100 + 100 + 494 * slide, 100 + 100 + 495 * slide, 12000):
        b, o, _ = synth.make_reads(1, L + int(rng.integers(0, 6)), motif, seed=int(rng.integers(1 << 30)), tract_min=500, tract_max=min(L, 4000))
        seqs.append(synth.split_reads(b, o)[0])
    prm = hiplib.make_params(min_len=0, min_count=0, slide=slide, maxlen=1 << 20, flags=hiplib.F_STEP1 | hiplib.F_WINDOWS | hiplib.F_STORE_SUMS | hiplib.F_STORE_RAW)
    emu.KNOBS.update(raw_m=2, val_off=1)
    try:
        L = emu.lib()
        t0 = L.emu_counter(0)
        out = emu.scan(pats, seqs, prm)
        assert L.emu_counter(0) > t0
    finally:
        emu.KNOBS.update(raw_m=0, val_off=0)
    rows = out["raw"]
    at = 0
    for i, seq in enumerate(seqs):
        tail = ["forward", "reverse"][int(out["results"]["tail"][i])]
        _, want = orc.window_count_matrix(seq, tail, pats, 100, 2 * slide, 100, 1 << 20)
        _, base = orc.window_count_matrix(seq, tail, pats, 100, slide, 100, 1 << 20)
        assert len(want) == (len(base) + 1) // 2
        assert np.array_equal(rows[at:at + len(want)], want), i
        assert np.array_equal(out["sums"][out["win_off"][i]:out["win_off"][i + 1]], base.sum(axis=1)), i
        at += len(want)


def test_planner_strided_scans_for_multiples_of_a_fused_slide():
    """Round 5 (second half): a slide without a fused kernel that is a multiple of one with -- raw rows or a self-overlap
    table at slide 10, 12, 14 ...; any table at 14, 15, 16, 18, 20 ... -- runs the base slide's fused kernel and keeps every m-th window."""
    def base(motif, k, slide, flags=0, window=100):
        return emu.stride_base(orc.kmer_table(motif, k), hiplib.make_params(slide=slide, window=window, flags=hiplib.F_STEP1 | hiplib.F_WINDOWS | hiplib.F_BINSEG | flags))
    raw = hiplib.F_STORE_RAW
    assert base("CCCTAA", 4, 10, raw) == 5 and base("CCCTAA", 4, 12, raw) == 6 and base("CCCTAA", 4, 14, raw) == 7 and base("CCCTAA", 4, 16, raw) == 8
    assert base("CCCTAA", 4, 15, raw) == 5 and base("CCCTAA", 4, 18, raw) == 6 and base("CCCTAA", 4, 9, raw) == 0 and base("CCCTAA", 4, 11, raw) == 0
    assert base("CCCTAA", 6, 10) == 5 and base("CCCTAA", 6, 12, raw) == 6 and base("CCCTAA", 5, 21) == 7       # self-overlap tables
    assert base("CCCTAA", 4, 10) == 0 and base("CCCTAA", 4, 6) == 0 and base("CCCTAA", 4, 6, raw) == 0        # a fused kernel of their own
    assert base("CCCTAA", 4, 20) == 10 and base("CCCTAA", 4, 24) == 12 and base("CCCTAA", 4, 22) == 11 and base("CCCTAA", 4, 13) == 0   # default tables: the largest base
    assert base("CCCTAA", 4, 49) == 0                                                                           # 7 x 7: more than 6 windows dropped per window kept
    assert base("CCCTAACCTA", 8, 10) == 0                                                                       # hashed table: generic at any slide


def test_planner_full_tiles_everywhere():
    """With the packed batch a tile is staged as whole 64-base quads whatever the slide, so every fused tile uses all 64
    lanes: 512 - q - 1 windows rounded down to an even number (494 at slide 6, window 100: tiles start at even windows because
    the 16-bit window sums leave in whole dwords), for the sums-only, raw-count and self-overlap kernels alike."""
    def plan(motif, k, slide, nwin, flags=0):
        return emu.plan_table(orc.kmer_table(motif, k), hiplib.make_params(slide=slide, flags=hiplib.F_WINDOWS | flags), nwin)
    for nwin in (2467, 3301):
        p = plan("CCCTAA", 4, 6, nwin)
        assert (p["tile_full"], p["tw"]) == (1, 494)
    assert plan("CCCTAA", 4, 6, 2467, hiplib.F_STORE_RAW)["tw"] == 494 and plan("CCCTAA", 5, 6, 2467)["tw"] == 496
    assert plan("AAACCCT", 5, 7, 2829)["tile_full"] == 1 and plan("TTAGG", 4, 5, 3000)["tile_full"] == 1


@pytest.mark.parametrize("pats,W", [(["AC", "GT"], 600), (["ACGT", "TGCA"], 1100), (["AAAA", "CCCC"], 300)])
def test_emulation_wide_windows_raw_capacity(pats, W):
    """Raw rows (and their accumulators) are bytes: a window in which a pattern can occur more than 255 times is refused
    with TPS_E_CAPACITY when raw counts are requested (it used to carry into the neighbouring pattern's byte); the same
    scan without raw counts stays exact."""
    rng = np.random.default_rng(W)
    unit = pats[0]
    seqs = [(unit * 900)[: 2 * W + 37], "".join("ACGT"[x] for x in rng.integers(0, 4, 3 * W)) + unit * 400]
    prm = hiplib.make_params(window=W, slide=6, trimfirst=0, maxlen=20000,
                             flags=hiplib.F_WINDOWS | hiplib.F_BINSEG | hiplib.F_TAILS_IN | hiplib.F_STORE_SUMS | hiplib.F_STORE_RAW)
    with pytest.raises(RuntimeError, match="raw counts are bytes"):
        emu.scan(pats, seqs, prm, tails=[0, 1])
    prm.flags &= ~hiplib.F_STORE_RAW
    out = emu.scan(pats, seqs, prm, tails=[0, 1])
    for i, seq in enumerate(seqs):
        _, counts = orc.window_count_matrix(seq, ["forward", "reverse"][i], pats, W, 6, 0, 20000)
        lo, hi = out["win_off"][i], out["win_off"][i + 1]
        assert np.array_equal(out["sums"][lo:hi], counts.sum(axis=1))


def test_planner_refuses_sums_beyond_32_bits():
    prm = hiplib.make_params(window=60000, slide=1, trimfirst=0, maxlen=600000, flags=hiplib.F_WINDOWS | hiplib.F_BINSEG)
    with pytest.raises(RuntimeError, match="exceed"):
        emu.plan(4, 12, prm, 500000)
    prm = hiplib.make_params(window=255 * 4 + 4, slide=6, flags=hiplib.F_WINDOWS | hiplib.F_STORE_RAW)
    emu.plan(4, 12, prm, 3000)                       # 255 occurrences at most: still fine


@pytest.mark.parametrize("motif,k,slide,units", [
    ("CCCTAA", 5, 6, ["CTAA", "GATT"]),
    ("CCCTAA", 6, 6, ["CCTAA", "GGATT", "CTAAC"]),
    ("CCCTAA", 6, 7, ["CCTAA"]),
    ("TTTAGGG", 7, 8, ["TTTAGG"]),
])
def test_emulation_self_overlap_sums_clean_batch_layout(motif, k, slide, units, monkeypatch):
    """The sums-only kernels of self-overlap tables on a batch WITHOUT non-ACGT letters (round 4): 16-bit pattern masks in the
    LDS table (ScanArgs::lut16) and XT aliased onto the head of the staged bases (xt_alias without xt_own) -- the layout the
    planner picks for a clean batch on the device.  Whole pipeline (step 1 on the 16-bit table, chain-corrected tiles, change
    point) against the oracle, both tails decided by step 1."""
    monkeypatch.setitem(emu.KNOBS, "val_off", 1)
    rng = np.random.default_rng(99 + 10 * k + slide)
    pats, seqs = _pp_reads(rng, motif, k, 6, 7000, units)
    seqs = [s if i % 2 == 0 else s[::-1] for i, s in enumerate(seqs)]
    plan = emu.plan_table(pats, hiplib.make_params(window=100, slide=slide), 3301)
    assert plan["variant"] == slide and plan["pp_d"] > 0
    L = emu.lib()
    flags = hiplib.F_STEP1 | hiplib.F_WINDOWS | hiplib.F_BINSEG | hiplib.F_STORE_SUMS
    prm = hiplib.make_params(no_bp=1000, min_len=1000, min_count=0, window=100, slide=slide, trimfirst=100, maxlen=20000, flags=flags)
    t0 = L.emu_counter(5) + L.emu_counter(6)
    out = emu.scan(pats, seqs, prm, base_shift=int(rng.integers(16)))
    assert L.emu_counter(5) + L.emu_counter(6) - t0 >= 6
    for i, seq in enumerate(seqs):
        cs, ce = orc.trc_counts(seq, pats)
        assert np.array_equal(out["c_start"][i], cs) and np.array_equal(out["c_end"][i], ce), i
        call = orc.trc_call(cs, ce, pats, len(motif), -1.0)
        tail = call[1]
        assert out["results"][i]["tail"] == (0 if tail == "forward" else 1)
        _, counts = orc.window_count_matrix(seq, tail, pats, 100, slide, 100, 20000)
        lo, hi = out["win_off"][i], out["win_off"][i + 1]
        assert np.array_equal(out["sums"][lo:hi], counts.sum(axis=1)), i
        assert out["results"][i]["bkp"] == orc.binseg_l2_exact(counts.sum(axis=1))


@pytest.mark.parametrize("window,fast", [(261, True), (262, False), (420, False)])
def test_emulation_self_overlap_sums_wide_windows(window, fast):
    """The chain-corrected sums tiles keep a window's matches and pairs in two 8-bit fields of one 16-bit count (round 4): windows
    of more than 255 start positions (W - k > 255) take the flag-and-recount tile instead -- both exact."""
    motif, k, slide = "CCCTAA", 6, 6
    rng = np.random.default_rng(window)
    pats, seqs = _pp_reads(rng, motif, k, 3, 8000, ["CCTAA", "GGATT"])
    prm = hiplib.make_params(window=window, slide=slide, trimfirst=100, maxlen=20000,
                             flags=hiplib.F_WINDOWS | hiplib.F_BINSEG | hiplib.F_TAILS_IN | hiplib.F_STORE_SUMS)
    L = emu.lib()
    t0 = L.emu_counter(5) + L.emu_counter(6)
    out = emu.scan(pats, seqs, prm, tails=[0, 1, 0])
    assert (L.emu_counter(5) + L.emu_counter(6) - t0 > 0) == fast
    for i, seq in enumerate(seqs):
        _, counts = orc.window_count_matrix(seq, ["forward", "reverse", "forward"][i], pats, window, slide, 100, 20000)
        lo, hi = out["win_off"][i], out["win_off"][i + 1]
        assert np.array_equal(out["sums"][lo:hi], counts.sum(axis=1)), i


@pytest.mark.parametrize("motif,k,unit", [
    ("CCCTAA", 6, "CCTAA"), ("CCCTAA", 5, "CTAA"), ("TTTAGGG", 7, "TTTAGG"), ("TTAGGG", 5, "TTAG"), ("CCCTAA", 6, "GGATT"),
])
def test_emulation_step1_chains_of_every_length(motif, k, unit):
    """Step 1 on tables with self-overlapping k-mers (round 4: pairs and triples taken off / added back per pattern from the packed
    bases, chains of four or more recounted): chains of 2 .. 9 elements planted at random places of both heads, across the 16-base
    chunk boundaries and across the END of the 1000-base head (where only the elements that are start positions of the head
    count), inside telomere-like repeats with deletions -- per-pattern counts of both ends against the oracle."""
    rng = np.random.default_rng(1000 * k + len(unit))
    pats = orc.kmer_table(motif, k)
    d = len(unit)
    seqs = []
    for i in range(48):
        L = int(rng.integers(1100, 2600)) if i % 4 else int(rng.integers(700, 1000))     # (short reads: the two heads overlap)
        if i % 3 == 0:
            body = list((motif * (L // len(motif) + 2))[:L])
            for _ in range(int(rng.integers(5, 60))):                       # deletions make pairs, close ones chains of three
                del body[int(rng.integers(0, len(body)))]
            body += ["ACGT"[x] for x in rng.integers(0, 4, L - len(body))]
        else:
            body = ["ACGT"[x] for x in rng.integers(0, 4, L)]
        for _ in range(int(rng.integers(1, 7))):
            n = int(rng.integers(2, 10))
            run = list(unit * (n - 1) + unit[:k])                           # n chained occurrences of the unit's k-mer(s)
            at = int(rng.choice([rng.integers(0, 40), rng.integers(950, 1010), rng.integers(0, L - len(run)), L - int(rng.integers(950, 1010)), L - len(run) - int(rng.integers(0, 30))]))
            at = max(0, min(L - len(run), at))
            body[at:at + len(run)] = run
        seqs.append("".join(body[:L]))
    prm = hiplib.make_params(no_bp=1000, min_len=0, min_count=10 ** 6, flags=hiplib.F_STEP1)
    out = emu.scan(pats, seqs, prm, base_shift=int(rng.integers(16)))
    for i, seq in enumerate(seqs):
        cs, ce = orc.trc_counts(seq, pats)
        assert np.array_equal(out["c_start"][i], cs), (i, len(seq), out["c_start"][i], cs)
        assert np.array_equal(out["c_end"][i], ce), (i, len(seq), out["c_end"][i], ce)


@pytest.mark.parametrize("motif,k,slide,units", [
    ("CCCTAA", 4, 6, []),                          # 12-byte rows, no self-overlap
    ("CCCTAA", 6, 6, ["CCTAA", "GGATT"]),          # 12-byte rows, period 5
    ("AAACCCT", 5, 7, []),                         # 14-byte rows: padded to 16 bytes in LDS, 18 lanes per pass
    ("TTAGGG", 5, 5, ["TTAG"]),
])
def test_emulation_raw_rows_clean_batch_layout(motif, k, slide, units, monkeypatch):
    """The raw-row kernels on a batch without non-ACGT letters (round 4): no XF / XT words in the exchange region -- the per-pattern
    tiles keep their lane totals in the pad words of END, the row staging buffer is XPC alone -- the LDS layout the planner picks
    for a clean batch on the device (xt_alias = 2 without xt_own).  Rows, sums and change point against the oracle."""
    monkeypatch.setitem(emu.KNOBS, "val_off", 1)
    rng = np.random.default_rng(7 + 10 * k + slide)
    pats, seqs = _pp_reads(rng, motif, k, 4, 8000, units)
    prm = hiplib.make_params(window=100, slide=slide, trimfirst=100, maxlen=20000,
                             flags=hiplib.F_WINDOWS | hiplib.F_BINSEG | hiplib.F_TAILS_IN | hiplib.F_STORE_SUMS | hiplib.F_STORE_RAW)
    L = emu.lib()
    t0 = L.emu_counter(0)
    tails = [0, 1, 1, 0]
    out = emu.scan(pats, seqs, prm, tails=tails, base_shift=int(rng.integers(16)))
    assert L.emu_counter(0) - t0 >= 8
    for i, seq in enumerate(seqs):
        _, counts = orc.window_count_matrix(seq, ["forward", "reverse"][tails[i]], pats, 100, slide, 100, 20000)
        lo, hi = out["win_off"][i], out["win_off"][i + 1]
        assert np.array_equal(out["raw"][lo:hi], counts), i
        assert np.array_equal(out["sums"][lo:hi], counts.sum(axis=1)), i
        assert out["results"][i]["bkp"] == orc.binseg_l2_exact(counts.sum(axis=1))


@pytest.mark.parametrize("motif,slide,W", [("AAACCCT", 7, 100), ("CCCTAAA", 6, 100), ("AAACCCT", 5, 100), ("CCCTAAA", 8, 260), ("TTTAGGG", 6, 101)])
def test_emulation_k5_pair_table_of_16_bit_masks(motif, slide, W, monkeypatch):
    """Round 4: k = 5 tables without self-overlap (the plant-type 7-mer motifs at the reference's default k) take two positions per
    lookup from a 16-bit pair table (ScanArgs::pair16, kernels _s*q; the single table in the same format).  On the device the planner
    picks it by itself (8-wave workgroups); the emulation's slices are bigger, so the test forces it.  Step 1, window sums and the
    boundary against the oracle -- reads of several tiles, both strands, a read with N, lower case."""
    monkeypatch.setitem(emu.KNOBS, "force_pair", 1)
    k = 5
    pats = orc.kmer_table(motif, k)
    prm = hiplib.make_params(no_bp=1000, min_len=0, min_count=-1, window=W, slide=slide, trimfirst=100, maxlen=20000,
                             flags=hiplib.F_STEP1 | hiplib.F_WINDOWS | hiplib.F_BINSEG | hiplib.F_STORE_SUMS)
    pl = emu.plan_table(pats, prm, 3000)
    assert pl["variant"] == slide and pl["pair_n"] == 4 ** (k + 1) // 2, pl
    rng = np.random.default_rng(slide * 13 + W)
    comp = str.maketrans("ACGTacgt", "TGCAtgca")
    seqs = []
    for i in range(5):
        L = int(rng.integers(4000, 9000))
        tract = int(rng.integers(600, 3500))
        body = list((motif * (tract // len(motif) + 2))[:tract] + "".join("ACGT"[x] for x in rng.integers(0, 4, L - tract)))
        for p in rng.integers(0, L, L // 14):
            body[p] = "ACGT"[int(rng.integers(4))]
        if i == 3:
            body[int(rng.integers(1200, L))] = "N"
        if i == 4:
            body[50:400] = [c.lower() for c in body[50:400]]
        sq = "".join(body)
        seqs.append(sq if i % 2 == 0 else sq[::-1].translate(comp))
    out = emu.scan(pats, seqs, prm)
    for i, seq in enumerate(seqs):
        cs, ce = orc.trc_counts(seq, pats, 1000)
        assert out["c_start"][i].tolist() == cs and out["c_end"][i].tolist() == ce, i
        r = out["results"][i]
        tail = "forward" if r["tail"] == 0 else "reverse"
        _, counts = orc.window_count_matrix(seq, tail, pats, W, slide, 100, 20000)
        lo, hi = out["win_off"][i], out["win_off"][i + 1]
        assert hi - lo == counts.shape[0] and hi - lo > 500
        assert np.array_equal(out["sums"][lo:hi], counts.sum(axis=1)), i
        assert r["bkp"] == orc.binseg_l2_exact(counts.sum(axis=1)), i


@pytest.mark.parametrize("k,slide", [(5, 6), (6, 6), (5, 5), (6, 7), (5, 7)])
def test_emulation_self_overlap_sums_blocks_with_a_pattern_twice(k, slide):
    """The sums tiles of the self-overlap tables count a block's matches as popcount(OR of its entries) + the few pairs of positions
    that can hold the SAME pattern (CD apart, or the block's first and last position).  Reads built to put a pattern twice into a
    block in every such way: runs of one k-mer repeated back to back (occurrences k apart: adjacent, not overlapping), (CCTAA)n /
    (CCTA)n runs (occurrences CD apart: the chains), telomeric stretches with deletions, all at every phase to the blocks."""
    motif = "CCCTAA"
    pats = orc.kmer_table(motif, k)
    rng = np.random.default_rng(k * 10 + slide)
    seqs = []
    for r in range(6):
        parts = ["".join("ACGT"[x] for x in rng.integers(0, 4, 150 + r))]
        for rep in range(40):
            kind = int(rng.integers(4))
            if kind == 0:
                p = pats[int(rng.integers(len(pats)))]
                parts.append(p * int(rng.integers(2, 9)))                     # the same k-mer k apart
            elif kind == 1:
                unit = motif[1:] if k == 6 else motif[2:]                     # CCTAA (period 5) / CTAA-type period-4 unit
                parts.append((unit * 12)[: int(rng.integers(k + 2, 50))])
            elif kind == 2:
                t = list(motif * int(rng.integers(3, 12)))
                for _ in range(int(rng.integers(0, 4))):
                    del t[int(rng.integers(len(t)))]
                parts.append("".join(t))
            else:
                parts.append("".join("ACGT"[x] for x in rng.integers(0, 4, int(rng.integers(1, 40)))))
        seqs.append("".join(parts))
    prm = hiplib.make_params(window=100, slide=slide, trimfirst=100, maxlen=20000,
                             flags=hiplib.F_WINDOWS | hiplib.F_BINSEG | hiplib.F_TAILS_IN | hiplib.F_STORE_SUMS)
    out = emu.scan(pats, seqs, prm, tails=[0] * len(seqs))
    for i, seq in enumerate(seqs):
        _, counts = orc.window_count_matrix(seq, "forward", pats, 100, slide, 100, 20000)
        lo, hi = out["win_off"][i], out["win_off"][i + 1]
        assert hi - lo == counts.shape[0] > 50
        assert np.array_equal(out["sums"][lo:hi], counts.sum(axis=1)), (k, slide, i)


@pytest.mark.parametrize("slide", [3, 4, 9, 10, 11, 12])
@pytest.mark.parametrize("force_pair", [False, True])
def test_emulation_default_kernels_other_slides(slide, force_pair, monkeypatch):
    """Round 4: the default kernels (sums only, no self-overlapping k-mer) are instantiated for slides 4 and 9 .. 12 as well -- those
    took the generic kernel before, three to five times slower per window.  Step 1, window sums and the boundary against the oracle;
    with and without the pair table (the emulation's planner drops it where its bigger slices make it cost a workgroup)."""
    if force_pair:
        monkeypatch.setitem(emu.KNOBS, "force_pair", 1)
    motif, k, W = "CCCTAA", 4, 100
    pats = orc.kmer_table(motif, k)
    prm = hiplib.make_params(no_bp=1000, min_len=0, min_count=-1, window=W, slide=slide, trimfirst=100, maxlen=20000,
                             flags=hiplib.F_STEP1 | hiplib.F_WINDOWS | hiplib.F_BINSEG | hiplib.F_STORE_SUMS)
    pl = emu.plan_table(pats, prm, 5000)
    assert pl["variant"] == slide and (pl["pair_n"] == 1024 or not force_pair), pl
    rng = np.random.default_rng(slide * 7 + force_pair)
    comp = str.maketrans("ACGTacgt", "TGCAtgca")
    seqs = []
    for i in range(5):
        L = int(rng.integers(6000, 15000))
        tract = int(rng.integers(800, 5000))
        body = list((motif * (tract // len(motif) + 2))[:tract] + "".join("ACGT"[x] for x in rng.integers(0, 4, L - tract)))
        for p in rng.integers(0, L, L // 12):
            body[p] = "ACGT"[int(rng.integers(4))]
        if i == 3:
            body[int(rng.integers(1500, L))] = "N"
        sq = "".join(body)
        seqs.append(sq if i % 2 == 0 else sq[::-1].translate(comp))
    out = emu.scan(pats, seqs, prm)
    for i, seq in enumerate(seqs):
        cs, ce = orc.trc_counts(seq, pats, 1000)
        assert out["c_start"][i].tolist() == cs and out["c_end"][i].tolist() == ce, i
        r = out["results"][i]
        tail = "forward" if r["tail"] == 0 else "reverse"
        _, counts = orc.window_count_matrix(seq, tail, pats, W, slide, 100, 20000)
        lo, hi = out["win_off"][i], out["win_off"][i + 1]
        assert hi - lo == counts.shape[0] and hi - lo > 400
        assert np.array_equal(out["sums"][lo:hi], counts.sum(axis=1)), i
        assert r["bkp"] == orc.binseg_l2_exact(counts.sum(axis=1)), i


@pytest.mark.parametrize("slide,W,force_pair", [(6, 100, True), (6, 100, False), (8, 100, True), (6, 102, True), (6, 101, True), (7, 100, True), (5, 100, True)])
def test_emulation_raw_rows_pair_table_of_fields(slide, W, force_pair, monkeypatch):
    """Round 5: the raw-row kernels of tables without self-overlap take two positions per lookup at k <= 4 from a pair table of one-hot
    FIELDS (tile_pp_s<.., PAIRF>; ScanArgs::pair_n with lut_fields) when the window's partial block ends between two pairs -- k = 4 at the
    default window, slide 6 -- and the partial-block position is a compile-time constant there (RPT).  Other windows / slides
    keep single lookups (the planner must not grant the table).  Rows, sums and the boundary against the oracle, reads of several
    tiles, both tails, a read with an N (its tile takes the fallback, which must not touch the fields table), lower case."""
    if force_pair:
        monkeypatch.setitem(emu.KNOBS, "force_pair", 1)
    motif, k = "CCCTAA", 4
    pats = orc.kmer_table(motif, k)
    prm = hiplib.make_params(window=W, slide=slide, trimfirst=100, maxlen=20000,
                             flags=hiplib.F_WINDOWS | hiplib.F_BINSEG | hiplib.F_TAILS_IN | hiplib.F_STORE_SUMS | hiplib.F_STORE_RAW)
    pl = emu.plan_table(pats, prm, 3000)
    assert pl["variant"] == slide, pl
    if force_pair:                             # (without the knob the occupancy rule decides: the emulation's slices are bigger than the device's)
        assert (pl["pair_n"] == 1024) == (slide == 6 and W == 100), pl
    else:
        assert pl["pair_n"] in (0, 1024), pl
    rng = np.random.default_rng(100 * slide + W)
    _, seqs = _pp_reads(rng, motif, k, 4, 9500, [])
    seqs[1] = seqs[1][:5000] + "N" + seqs[1][5001:]
    seqs[2] = seqs[2][:300] + seqs[2][300:900].lower() + seqs[2][900:]
    tails = [0, 1, 0, 1]
    L = emu.lib()
    t0 = L.emu_counter(0)
    out = emu.scan(pats, seqs, prm, tails=tails, base_shift=int(rng.integers(16)))
    assert L.emu_counter(0) - t0 >= (8 if slide != 8 else 0)        # (slide 8: a lane could hold 16 occurrences of a 4-mer -- no per-pattern tile, the fallback counts)
    for i, seq in enumerate(seqs):
        _, counts = orc.window_count_matrix(seq, ["forward", "reverse"][tails[i]], pats, W, slide, 100, 20000)
        lo, hi = out["win_off"][i], out["win_off"][i + 1]
        assert np.array_equal(out["raw"][lo:hi], counts), i
        assert np.array_equal(out["sums"][lo:hi], counts.sum(axis=1)), i
        assert out["results"][i]["bkp"] == orc.binseg_l2_exact(counts.sum(axis=1))


@pytest.mark.parametrize("k,W", [(5, 100), (6, 100), (5, 103), (6, 97)])
def test_emulation_self_overlap_raw_rows_home_and_other_windows(k, W):
    """Round 5: the self-overlap raw-row tiles carry the partial-block position of their home shape (CCCTAA at k = 5 / 6, window 100) as a
    compile-time constant; other windows take the run-time instantiation.  Both against the oracle."""
    motif, slide = "CCCTAA", 6
    rng = np.random.default_rng(31 * k + W)
    pats, seqs = _pp_reads(rng, motif, k, 3, 9000, ["CTAA", "GATT"] if k == 5 else ["CCTAA", "GGATT"])
    prm = hiplib.make_params(window=W, slide=slide, trimfirst=100, maxlen=20000,
                             flags=hiplib.F_WINDOWS | hiplib.F_BINSEG | hiplib.F_TAILS_IN | hiplib.F_STORE_SUMS | hiplib.F_STORE_RAW)
    tails = [0, 1, 1]
    out = emu.scan(pats, seqs, prm, tails=tails, base_shift=int(rng.integers(16)))
    for i, seq in enumerate(seqs):
        _, counts = orc.window_count_matrix(seq, ["forward", "reverse"][tails[i]], pats, W, slide, 100, 20000)
        lo, hi = out["win_off"][i], out["win_off"][i + 1]
        assert np.array_equal(out["raw"][lo:hi], counts), i
        assert np.array_equal(out["sums"][lo:hi], counts.sum(axis=1)), i


@pytest.mark.parametrize("motif,k,slide", [("TTTTAGGG", 6, 8), ("TTTTAGGG", 6, 6), ("CCCTAAAA", 6, 8), ("TTTTAGGG", 5, 8), ("AACCGGTT", 6, 8)])
def test_emulation_sixteen_patterns_take_the_fused_tiles(motif, k, slide, monkeypatch):
    """Round 5: an 8-letter motif has P = 16 k-mers (+ complements) at the reference's default k = len - 2 -- one more than the fused
    kernels' 16-bit masks held while bit 15 was reserved for the fallback tile's self-overlap flag.  A table WITHOUT self-overlapping
    k-mers has no use for the flag: it takes the default tiles (sums only); step 1 counts through the histogram path (the packed
    counter's squaring trick needs bit 15 free).  Step 1, sums, boundary against the oracle; reads of several tiles, both strands, N."""
    pats = orc.kmer_table(motif, k)
    assert len(pats) == 16
    so = any(p[:d] == p[-d:] for p in pats for d in range(1, k))           # TTTTAGGG at k = 6: TAGGGT / ATCCCA have period 5
    prm = hiplib.make_params(no_bp=1000, min_len=0, min_count=-1, window=100, slide=slide, trimfirst=100, maxlen=20000,
                             flags=hiplib.F_STEP1 | hiplib.F_WINDOWS | hiplib.F_BINSEG | hiplib.F_STORE_SUMS)
    if so:
        # a self-overlap table of sixteen: the chain-corrected tiles on a batch without non-ACGT letters (the fallback tile's flag bit
        # is the sixteenth pattern); a batch WITH such letters keeps the generic kernel
        assert emu.plan_table(pats, prm, 3000)["variant"] == 0
        monkeypatch.setitem(emu.KNOBS, "val_off", 1)
    pl = emu.plan_table(pats, prm, 3000)
    assert pl["variant"] == slide, pl
    rng = np.random.default_rng(slide + 17 * k)
    comp = str.maketrans("ACGTacgt", "TGCAtgca")
    seqs = []
    for i in range(4):
        tract = (motif * 500)[int(rng.integers(0, 8)):][: int(rng.integers(800, 3000))]
        body = list(tract) + ["ACGT"[x] for x in rng.integers(0, 4, 9000 - len(tract))]
        for _ in range(60):
            p = int(rng.integers(0, len(tract)))
            body[p] = "ACGT"[int(rng.integers(4))]
        s = "".join(body)
        if i == 2 and not so:
            s = s[:2500] + "N" + s[2501:]
        if i & 1:
            s = s.translate(comp)[::-1]
        seqs.append(s)
    if so:                                       # overlapping occurrences of TAGGGT: (TAGGG)n runs and deletion-made pairs
        seqs[0] = seqs[0][:3000] + "TAGGG" * 9 + "T" + seqs[0][3046:]
        seqs[3] = seqs[3][:1500] + "TTTTAGGGTAGGGTTTTAGGG" + seqs[3][1521:]
    out = emu.scan(pats, seqs, prm)
    for i, seq in enumerate(seqs):
        cs, ce = orc.trc_counts(seq, pats)
        assert np.array_equal(out["c_start"][i], cs) and np.array_equal(out["c_end"][i], ce), i
        tail = ["forward", "reverse"][int(out["results"][i]["tail"])]
        _, counts = orc.window_count_matrix(seq, tail, pats, 100, slide, 100, 20000)
        lo, hi = out["win_off"][i], out["win_off"][i + 1]
        assert np.array_equal(out["sums"][lo:hi], counts.sum(axis=1)), i
        assert out["results"][i]["bkp"] == orc.binseg_l2_exact(counts.sum(axis=1))
    # raw rows of 16 patterns keep the generic kernel (rows of at most 14 bytes in the per-pattern tiles)
    prm_raw = hiplib.make_params(window=100, slide=slide, trimfirst=100, maxlen=20000, flags=hiplib.F_WINDOWS | hiplib.F_TAILS_IN | hiplib.F_STORE_SUMS | hiplib.F_STORE_RAW)
    assert emu.plan_table(pats, prm_raw, 3000)["variant"] == 0
