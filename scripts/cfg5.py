import sys, time, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
import numpy as np
from topsicle_amd import hiplib, synth, allsteps
b,o,_ = synth.make_reads(4000, 25000, "CCCTAA", 5)
sc = hiplib.HipScanner(0)
for k in (4,5,6):
    pats = allsteps.patterns_to_search("CCCTAA", k); sc.set_patterns(pats)
    sc.upload(0, b, o)
    for raw in (0,1):
        prm = hiplib.make_params(min_len=9000, min_count=allsteps.min_count_for_cutoff(0.7, 1000/6, 1000), slide=6, flags=1|2|4|8|(16 if raw else 0))
        sc.scan(0, prm); sc.sync(); sc.kernel_time_reset()
        for i in range(3): sc.scan(0, prm)
        sc.sync()
        n, tot, mean = sc.kernel_time_ms()
        print(f"k={k} raw={raw}: kernel {mean:.3f} ms  -> {b.size/mean/1e6:.1f} G bases/s", flush=True)
