"""Randomised check of the several-tables-at-once path (run on the GPU box): random motifs, 2-4 values of k, random window
parameters and output flags, batch after batch of random size through batch.scan_jobs -- helper contexts borrowing the
resident batch, all tables launched together -- against the same jobs back to back on the one context
(batch.SEQUENTIAL_TABLES).  Rows, window sums and raw rows of passing reads must be identical."""
import os, sys, time
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT)
import numpy as np
from topsicle_amd import allsteps, batch, hiplib, synth

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
sc = hiplib.HipScanner(0)
motifs = ["CCCTAA", "AAACCCT", "TTAGGG", "TTAGG", "CCCTAAA", "TTTTGGGG"]
t0 = time.time()
bad = 0
for case in range(n_cases):
    rng = np.random.default_rng(seed0 * 100003 + case)
    motif = motifs[int(rng.integers(len(motifs)))]
    ks = sorted(set(int(x) for x in rng.integers(3, len(motif) + 1, int(rng.integers(2, 5)))))
    if len(ks) < 2:
        continue
    slide = int(rng.choice([6, 7, 5, 8, len(motif), 3, 10, 12, 14, 15, 16, 20]))      # (10 and up: no fused kernel of their own -- strided scans, tps::stride_base)
    W = int(rng.choice([100, 100, 60, 150]))
    prm = hiplib.make_params(no_bp=int(rng.choice([1000, 500])), min_len=int(rng.choice([0, 1000, 3000])),
                             min_count=int(rng.integers(0, 40)), window=W, slide=slide, trimfirst=int(rng.choice([100, 0])),
                             maxlen=int(rng.choice([20000, 6000])), jump=int(rng.choice([5, 1, 3])), min_size=2,
                             flags=hiplib.F_STEP1 | hiplib.F_WINDOWS | hiplib.F_BINSEG)
    jobs = []
    for k in ks:
        pats = allsteps.patterns_to_search(motif, k)
        if len(pats) > hiplib.MAX_PATTERNS:
            continue
        jobs.append(batch.Job(pats, prm, want_sums=bool(rng.integers(2)), want_raw=bool(rng.integers(2))))
    if len(jobs) < 2:
        continue
    for b in range(int(rng.integers(1, 4))):                      # several batches through the same contexts
        n = int(rng.integers(1, 400))
        bases, offsets = synth.make_ragged_reads(n, motif, int(rng.integers(1 << 30)), errors=synth.ONT, n_frac=0.0005, lower_frac=0.1,
                                                 len_mu=float(rng.choice([7.5, 8.5, 9.3])), len_sigma=0.8, min_len=30, max_len=40000)[:2]
        recs = type("B", (), {"bases": bases, "offsets": offsets})()
        try:
            batch.SEQUENTIAL_TABLES = True
            seq = batch.scan_jobs(sc, recs, jobs)
            batch.SEQUENTIAL_TABLES = False
            con = batch.scan_jobs(sc, recs, jobs)
        except hiplib.TopsicleHipError as e:
            print("case", case, "rejected:", str(e)[:100])
            break
        for j, ((r1, s1, w1, o1), (r2, s2, w2, o2)) in enumerate(zip(seq, con)):
            ok = all(np.array_equal(r1[f], r2[f]) for f in r1.dtype.names)
            if s1 is not None:
                keep = np.repeat(r1["pass"].astype(bool), np.diff(o1))
                ok = ok and np.array_equal(o1, o2) and np.array_equal(s1[keep], s2[keep])
            if w1 is not None:
                keep = np.repeat(r1["pass"].astype(bool), np.diff(o1))
                ok = ok and np.array_equal(w1[keep], w2[keep])
            if not ok:
                bad += 1
                print("MISMATCH case", case, "batch", b, "job", j, motif, ks, slide, W)
print(f"{n_cases} cases, {bad} mismatching table passes, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
