"""Randomised GPU-vs-oracle sweep (run on the GPU box; the C oracle is the checker, test infrastructure only).
Many (motif, k, window, slide, trimfirst, maxlen, jump) combinations x reads with errors / N / lower case / both
strands / odd lengths; step-1 counts, window sums, raw counts and the change-point must all agree."""
import os, sys, time
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
from topsicle_amd import hiplib, allsteps
import oracle_c, topsicle_oracle as orc
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
sc = hiplib.HipScanner(0)
motifs = ["CCCTAA", "AAACCCT", "TTAGGG", "TTAGG", "CCCTAAA", "TTTTGGGG", "CCCTAACCTA", "TTAGGC", "AACCCT", "CCCGAA"]
bad = 0
t0 = time.time()
for case in range(n_cases):
    rng = np.random.default_rng(seed0 * 100003 + case)
    motif = motifs[int(rng.integers(len(motifs)))]
    k = int(rng.integers(3, min(len(motif) + 2, 10)))
    pats = allsteps.patterns_to_search(motif, k)
    if len(pats) > hiplib.MAX_PATTERNS:
        continue
    W = int(rng.choice([100, 100, 100, 60, 150, 260, 37]))
    s = int(rng.choice([6, 7, 5, 8, len(motif), 3, 10, 13, 12, 14, 15, 16, 20, 21]))      # (10, 12, 14 ...: strided scans of a fused base slide for raw rows and self-overlap tables, for every table above 12)
    t = int(rng.choice([100, 0, 50]))
    M = int(rng.choice([20000, 4000, 12000]))
    jump = int(rng.choice([5, 5, 1, 3, 8]))
    min_size = int(rng.choice([2, 2, 1, 4]))
    no_bp = int(rng.choice([1000, 1000, 500, 1400, 300]))
    Lmax = int(rng.choice([14000, 14000, 3000, 32000]))
    seqs = []
    for i in range(int(rng.integers(4, 14))):
        L = int(rng.integers(0, Lmax))
        tract = int(rng.integers(0, max(1, min(L, 6000))))
        ph = int(rng.integers(len(motif)))
        body = list((motif * (tract // len(motif) + 2))[ph:ph + tract] + "".join("ACGT"[x] for x in rng.integers(0, 4, max(0, L - tract))))
        ne = int(len(body) * rng.choice([0.0, 0.01, 0.06]))
        for p in rng.integers(0, max(1, len(body)), ne):
            if not body:
                break
            p = min(int(p), len(body) - 1)
            r = rng.random()
            if r < 0.5: body[p] = "ACGT"[int(rng.integers(4))]
            elif r < 0.8 and len(body) > 1: del body[p]
            elif r < 0.95: body.insert(p, "ACGT"[int(rng.integers(4))])
            else: body[p] = "NnacgtRY-"[int(rng.integers(9))]
        q = "".join(body)
        if rng.random() < 0.5:
            q = q[::-1].translate(str.maketrans("ACGTacgt", "TGCAtgca"))
        seqs.append(q)
    if not any(seqs):
        continue
    sc.set_patterns(pats)
    bases, offsets = hiplib.pack_reads(seqs)
    sc.upload(0, bases, offsets)
    prm = hiplib.make_params(no_bp=no_bp, min_len=0, min_count=-1, window=W, slide=s, trimfirst=t, maxlen=M, jump=jump, min_size=min_size,
                             flags=hiplib.F_STEP1 | hiplib.F_WINDOWS | hiplib.F_BINSEG | hiplib.F_STORE_SUMS | (hiplib.F_STORE_RAW if case % 3 == 0 else 0))
    try:
        sc.scan(0, prm); sc.sync()
    except hiplib.TopsicleHipError as e:
        print("case", case, "rejected:", e); continue
    res = sc.results(0).copy()
    sums, win_off = sc.window_sums(0)
    raw = sc.window_raw(0)[0] if case % 3 == 0 else None
    cs_all, ce_all = sc.batch_trc_counts(0)
    # the standalone entry points (tps_trc_counts / tps_window_counts / tps_binseg_l2) must agree with the fused scan
    if case % 5 == 0:
        cs2, ce2 = sc.trc_counts(bases, offsets, no_bp)
        s2, wo2, _ = sc.window_counts(bases, offsets, res["tail"].astype(np.uint8), W, s, t, M)
        b2, _g2 = sc.binseg_l2(sums, win_off, len(pats), jump, min_size)
        # (binseg_l2 hands exact ties to ruptures' float64 arithmetic; so does resolve_ties for the fused scan's flagged reads)
        res_t = res.copy()
        hiplib.resolve_ties(sc, 0, res_t, len(pats), jump, min_size)
        if not (np.array_equal(cs2, cs_all) and np.array_equal(ce2, ce_all) and np.array_equal(wo2, win_off) and np.array_equal(s2, sums)
                and np.array_equal(b2, res_t["bkp"])):
            bad += 1
            print(f"MISMATCH case {case}: standalone entry points differ from the fused scan (motif {motif} k {k} W {W} s {s} jump {jump} min_size {min_size})")
    for i, q in enumerate(seqs):
        cs, ce = oracle_c.trc_counts(q, pats, no_bp)
        ok = cs_all[i].tolist() == list(cs) and ce_all[i].tolist() == list(ce)
        why = "" if ok else "step1"

        tail = ["forward", "reverse"][res["tail"][i]]
        counts = oracle_c.window_counts(q, tail, pats, W, s, t, M)[1].astype(np.int64)
        got = sums[win_off[i]:win_off[i + 1]]
        if ok and not (got.shape[0] == counts.shape[0] and np.array_equal(got, counts.sum(axis=1))):
            ok, why = False, f"sums (n {got.shape[0]} vs {counts.shape[0]})"
        if raw is not None and ok and not np.array_equal(raw[win_off[i]:win_off[i + 1]], counts):
            ok, why = False, "raw"
        if ok and counts.shape[0]:
            want = orc.binseg_l2_exact(counts.sum(axis=1), jump, min_size) if counts.shape[0] >= 1 else None
            if res["bkp"][i] != (-1 if want is None else want):
                ok, why = False, f"bkp {res['bkp'][i]} vs {want} (n_win {counts.shape[0]})"
        if not ok:
            bad += 1
            if bad <= 40:
                print(f"MISMATCH case {case} read {i} [{why}]: motif {motif} k {k} W {W} s {s} t {t} M {M} jump {jump} min_size {min_size} no_bp {no_bp} len {len(q)}")
print(f"{n_cases} cases, {bad} mismatching reads, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
