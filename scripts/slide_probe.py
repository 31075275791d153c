import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from topsicle_amd import hiplib, synth, allsteps
motif = "CCCTAA"
b, o, _ = synth.make_reads(10000, 15000, motif, 20250920)
sc = hiplib.HipScanner(0)
for k, slide, flags in ((4, 10, 31), (4, 5, 31), (4, 12, 31), (4, 6, 31), (6, 10, 15), (6, 5, 15), (6, 10, 31), (6, 5, 31), (4, 10, 15)):
    sc.set_patterns(allsteps.patterns_to_search(motif, k))
    prm = hiplib.make_params(min_len=9000, min_count=allsteps.min_count_for_cutoff(0.7, 1000 / len(motif), 1000), slide=slide, flags=flags)
    sc.upload(0, b, o)
    for _ in range(5): sc.scan(0, prm)
    sc.sync(); sc.kernel_time_reset()
    for _ in range(40): sc.scan(0, prm)
    sc.sync()
    cnt, ms, mean = sc.kernel_time_ms()
    print("k=%d slide=%2d flags=%d  %8.1f us per launch  %s" % (k, slide, flags, ms / cnt * 1e3, sc.kernel_info(0).split()[0]), flush=True)
