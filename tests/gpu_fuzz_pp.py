"""Adversarial GPU-vs-oracle sweep for the per-pattern tiles: telomere repeats full of deletions and long runs of the
k-mers' own period (chains of overlapping occurrences), raw counts and sums, both tails.  The C oracle is the checker
(test infrastructure only).  usage: gpu_fuzz_pp.py [N_CASES] [SEED0]"""
import os, sys, time
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from topsicle_amd import hiplib
import topsicle_oracle as orc
from test_emulation import _pp_reads
cases = [("CCCTAA", 5, 6, ["CTAA", "GATT"]), ("CCCTAA", 6, 6, ["CCTAA", "GGATT", "CTAAC"]), ("CCCTAA", 6, 5, ["CCTAA"]), ("CCCTAA", 6, 7, ["CCTAA", "GGATT"]),
         ("CCCTAA", 6, 8, ["CCTAA"]), ("TTTAGGG", 7, 6, ["TTTAGG", "AAATCC"]), ("TTAGGG", 5, 5, ["TTAG", "AATC"]), ("CCCTAA", 4, 6, []), ("AAACCCT", 5, 7, []),
         ("ACACGT", 4, 6, ["AC", "GT", "TG"]), ("TTAGGG", 6, 6, ["TTAGG"]), ("AAACCCT", 7, 8, ["AAACCC", "TTTGGG"]), ("AAACCCT", 6, 7, ["AACCCT"]),
         ("TTAGG", 5, 5, ["TTAG"]), ("TTTTGGGG", 6, 8, ["TTTTGGG", "GGGGTTT"])]
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
sc = hiplib.HipScanner(0)
bad = 0
t0 = time.time()
for case in range(n_cases):
    rng = np.random.default_rng(seed0 * 7919 + case)
    motif, k, slide, units = cases[case % len(cases)]
    W = int(rng.choice([100, 100, 100, 80, 120, 150]))
    t = int(rng.choice([100, 0, 37]))
    pats, seqs = _pp_reads(rng, motif, k, int(rng.integers(3, 12)), int(rng.integers(3000, 24000)), units)
    if case % 2:
        # mostly random sequence with short chain-ridden stretches: most tiles take the chain-free path, the stretches land
        # anywhere relative to the tile boundaries (the hand-over between the plain and the canonical-pick tiles)
        mixed = []
        for sq in seqs:
            out, pos = [], 0
            while pos < len(sq):
                n = int(rng.integers(100, 1500))
                out.append(sq[pos:pos + n]); pos += n
                out.append("".join("ACGT"[x] for x in rng.integers(0, 4, int(rng.integers(1500, 9000)))))
            mixed.append("".join(out)[:int(rng.integers(6000, 30000))])
        seqs = mixed
    if rng.random() < 0.3:
        i = int(rng.integers(len(seqs))); p = int(rng.integers(len(seqs[i])))
        seqs[i] = seqs[i][:p] + "N" + seqs[i][p + 1:]
    tails = [int(x) for x in rng.integers(0, 2, len(seqs))]
    sc.set_patterns(pats)
    bases, offsets = hiplib.pack_reads(seqs)
    try:
        sums, win_off, raw = sc.window_counts(bases, offsets, tails, W, slide, t, 20000, raw=True)
        sums2, _, _ = sc.window_counts(bases, offsets, tails, W, slide, t, 20000, raw=False)
    except hiplib.TopsicleHipError as e:
        print("case", case, "rejected:", e); continue
    for i, seq in enumerate(seqs):
        _, counts = orc.window_count_matrix(seq, ["forward", "reverse"][tails[i]], pats, W, slide, t, 20000)
        lo, hi = win_off[i], win_off[i + 1]
        ok = hi - lo == counts.shape[0] and np.array_equal(raw[lo:hi], counts.reshape(-1, len(pats))) and \
            np.array_equal(sums[lo:hi], counts.sum(axis=1)) and np.array_equal(sums2[lo:hi], counts.sum(axis=1))
        if not ok:
            bad += 1
            print("MISMATCH case", case, motif, k, slide, W, t, "read", i, "tail", tails[i], flush=True)
print(f"{n_cases} cases, {bad} mismatching reads, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
