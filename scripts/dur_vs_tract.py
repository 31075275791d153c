"""Diagnostic: a read's duration (clock stamps of the diagnostics build) against its telomere tract length and its phases.
usage: python3 scripts/dur_vs_tract.py [K [FLAGS [N [LEN]]]]"""
import os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT)
import numpy as np
from topsicle_amd import hiplib, synth, allsteps
motif, slide = "CCCTAA", 6
k = int(sys.argv[1]) if len(sys.argv) > 1 else 6
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 15
n = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
rl = int(sys.argv[4]) if len(sys.argv) > 4 else 25000
pats = allsteps.patterns_to_search(motif, k)
b, o, truth = synth.make_reads(n, rl, motif, 20250920)
prm = hiplib.make_params(min_len=9000, min_count=allsteps.min_count_for_cutoff(0.7, 1000 / len(motif), 1000), slide=slide, flags=flags)
sc = hiplib.HipScanner(0, lib_path=os.path.join(ROOT, "topsicle_amd", "libtopsicle_hip_diag.so"))
sc.set_patterns(pats)
sc.upload(0, b, o)
for _ in range(50):
    sc.scan(0, prm)
sc.sync()
sc.debug_option("stamps", 1)
sc.scan(0, prm); sc.sync()
st = sc.stamps(0).astype(np.int64)
res = sc.results(0)
print(sc.kernel_info(0))
dur = (st[:, 10] - st[:, 13]) / 2400.0
tract = truth["tract"]
passed = st[:, 5] != 0
print("reads %d, past the TRC filter %d; duration of the others: mean %.2f us" % (n, passed.sum(), dur[~passed].mean() if (~passed).any() else 0))
edges = np.arange(1000, 9000, 1000)
print("tract      reads   dur_us(mean p10 p90)   step1+decide  tile0(staged..done)  other tiles  result   [clocks]")
for lo in edges[:-1]:
    s = passed & (tract >= lo) & (tract < lo + 1000)
    if not s.any():
        continue
    x = st[s]
    print("%5d-%5d %6d   %6.2f %6.2f %6.2f   %9.0f  %9.0f  %9.0f  %7.0f" % (lo, lo + 1000, s.sum(), dur[s].mean(), np.percentile(dur[s], 10), np.percentile(dur[s], 90),
          (x[:, 4] - x[:, 13]).mean(), (x[:, 8] - x[:, 4]).mean(), (x[:, 9] - x[:, 8]).mean(), (x[:, 10] - x[:, 9]).mean()))
A = np.stack([np.ones(passed.sum()), tract[passed]], 1)
coef, *_ = np.linalg.lstsq(A, dur[passed], rcond=None)
print("least squares over the passing reads: duration = %.2f us + %.2f us per 1000 bases of tract" % (coef[0], coef[1] * 1000))
