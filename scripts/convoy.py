"""Diagnostic: does lock-step (all reads the same length) cost throughput?  Same total bases, fixed vs random lengths."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from topsicle_amd import hiplib, synth, allsteps
motif, k = "CCCTAA", 4
pats = allsteps.patterns_to_search(motif, k)
b, o, _ = synth.make_reads(10000, 15000, motif, 1)
prm = hiplib.make_params(min_len=9000, min_count=100, slide=6, flags=15)
sc = hiplib.HipScanner(0); sc.set_patterns(pats)
rng = np.random.default_rng(0)
def repack(lens):
    parts = [b[o[i]:o[i] + lens[i]] for i in range(len(lens))]
    off = np.zeros(len(lens) + 1, np.int64); off[1:] = np.cumsum(lens)
    return np.concatenate(parts), off
for name, lens in [("fixed 12500", np.full(10000, 12500)), ("uniform 10000..15000", rng.integers(10000, 15001, 10000)),
                   ("fixed 15000", np.full(10000, 15000)), ("sorted desc 10000..15000", np.sort(rng.integers(10000, 15001, 10000))[::-1])]:
    bb, oo = repack(lens)
    for s in range(4):
        sc.upload(s, bb, oo)
    for i in range(70):
        sc.scan(i % 4, prm)
    sc.sync(); sc.kernel_time_reset()
    for i in range(40):
        sc.scan(i % 4, prm)
    sc.sync()
    n, tot, mean = sc.kernel_time_ms()
    print(f"{name:28s} bases {oo[-1]:>10d}  kernel {mean:.4f} ms  {mean * 1e6 / oo[-1] * 1e3:.3f} ps/base")
