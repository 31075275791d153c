"""GPU check: raw rows for any number of patterns (odd row lengths leave byte by byte)."""
import os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from topsicle_amd import hiplib
import topsicle_oracle as orc
from test_emulation import _pp_reads
sc = hiplib.HipScanner(0)
bad = 0
for motif, k, slide, counts in (("CCCTAA", 5, 6, [1, 2, 3, 5, 6, 7, 9, 10, 11, 12]), ("AAACCCT", 5, 7, [13, 14]), ("CCCTAA", 4, 6, [3, 5, 7, 11, 12])):
    rng = np.random.default_rng(k * 100 + slide)
    table = orc.kmer_table(motif, k)
    _, seqs = _pp_reads(rng, motif, k, 5, 9000, [])
    tails = [0, 1, 0, 1, 0]
    bases, offsets = hiplib.pack_reads(seqs)
    for P in counts:
        pats = table[:P]
        sc.set_patterns(pats)
        sums, win_off, raw = sc.window_counts(bases, offsets, tails, 100, slide, 100, 20000, raw=True)
        for i, seq in enumerate(seqs):
            _, want = orc.window_count_matrix(seq, ["forward", "reverse"][tails[i]], pats, 100, slide, 100, 20000)
            lo, hi = win_off[i], win_off[i + 1]
            ok = np.array_equal(raw[lo:hi], want.reshape(-1, P)) and np.array_equal(sums[lo:hi], want.sum(axis=1))
            if not ok:
                bad += 1; print("MISMATCH", motif, k, slide, P, i)
print("odd-P raw rows:", "ok" if not bad else f"{bad} mismatches")
sys.exit(1 if bad else 0)
