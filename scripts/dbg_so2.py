import sys, os
sys.path[:0]=[os.environ.get('GRAFT_REPO_ROOT','/root/repo')]
import numpy as np
from topsicle_amd import hiplib, synth, allsteps
def run(k, withN, label):
    motif="CCCTAA"
    pats=allsteps.patterns_to_search(motif,k)
    bases, offsets, truth = synth.make_reads(2000, 6000, motif, seed=99, tract_min=500, tract_max=3000)
    b=bases.copy()
    if withN:
        rng=np.random.default_rng(1); pos=rng.integers(0,b.size,b.size//200); b[pos]=np.frombuffer(b"NnacgtRY",dtype=np.uint8)[rng.integers(0,8,pos.size)]
    sc=hiplib.HipScanner(0); sc.set_patterns(pats); sc.upload(0,b,offsets)
    prm=hiplib.make_params(min_len=1000,min_count=-1,flags=1|2|4|8)
    sc.scan(0,prm); sc.sync(); s1,wo=sc.window_sums(0)
    diffs=[]
    for rep in range(6):
        sc.scan(0,prm); sc.sync(); t1,_=sc.window_sums(0); diffs.append(int((s1!=t1).sum()))
    print(label, 'run-to-run diffs', diffs)
    sc.close()
run(5, False, 'k=5 (SO), no N   ')
run(4, True,  'k=4 (no SO), N   ')
run(5, True,  'k=5 (SO), N      ')
run(4, False, 'k=4, no N        ')
