import sys, time, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
import numpy as np
from topsicle_amd import hiplib, synth, allsteps
pats = allsteps.patterns_to_search("CCCTAA", 4)
b,o,_ = synth.make_reads(10000, 15000, "CCCTAA", 1)
sc = hiplib.HipScanner(0); sc.set_patterns(pats)
NS=8
for s in range(NS): sc.upload(s, b, o)
prm = hiplib.make_params(min_len=9000, min_count=116, flags=1|2|4|8)
for i in range(2*NS): sc.scan(i%NS, prm)
sc.sync()
for n in (1, 10, 40, 40):
    sc.kernel_time_reset()
    t0=time.perf_counter()
    for i in range(n): sc.scan(i%NS, prm)
    t1=time.perf_counter()
    sc.sync()
    t2=time.perf_counter()
    print(n, 'submit/step us %.1f' % ((t1-t0)/n*1e6), 'total/step us %.1f' % ((t2-t0)/n*1e6), 'kernel ms', sc.kernel_time_ms()[2])
