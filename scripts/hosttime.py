import sys, time, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
import numpy as np
from topsicle_amd import hiplib, synth, allsteps
pats = allsteps.patterns_to_search("CCCTAA", 4)
b,o,_ = synth.make_reads(10000, 15000, "CCCTAA", 1)
sc = hiplib.HipScanner(0); sc.set_patterns(pats)
for s in range(4): sc.upload(s, b, o)
prm = hiplib.make_params(min_len=9000, min_count=116, flags=1|2|4|8)
for i in range(5): sc.scan(i%4, prm)
sc.sync()
for n in (1, 10, 50):
    t0=time.perf_counter()
    for i in range(n): sc.scan(i%4, prm)
    t1=time.perf_counter()
    sc.sync()
    t2=time.perf_counter()
    print(n, 'submit per step us', (t1-t0)/n*1e6, 'total per step us', (t2-t0)/n*1e6)
# single-step latency
for i in range(3):
    t0=time.perf_counter(); sc.scan(0, prm); sc.sync(); print('one step us', (time.perf_counter()-t0)*1e6)
