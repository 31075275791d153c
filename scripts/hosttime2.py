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
for s in range(NS): sc.scan(s, prm)
sc.sync()
for i in range(5): sc.scan(i%NS, prm)
sc.sync()
sc.kernel_time_reset()
ts=[time.perf_counter()]
for i in range(40):
    sc.scan(i%NS, prm); ts.append(time.perf_counter())
sc.sync(); te=time.perf_counter()
d=np.diff(ts)*1e6
print('per-call us: min %.1f med %.1f max %.1f sum %.0f ; total/step %.1f' % (d.min(), np.median(d), d.max(), d.sum(), (te-ts[0])/40*1e6))
print(np.round(d,0).tolist())
