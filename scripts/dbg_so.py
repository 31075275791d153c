import sys, os
sys.path[:0]=[os.environ.get('GRAFT_REPO_ROOT','/root/repo'), os.path.join(os.environ.get('GRAFT_REPO_ROOT','/root/repo'),'oracle')]
import numpy as np, topsicle_oracle as orc
from topsicle_amd import hiplib, synth
motif,k="CCCTAA",5
pats=orc.kmer_table(motif,k)
bases, offsets, truth = synth.make_reads(2000, 6000, motif, seed=99, tract_min=500, tract_max=3000)
rng=np.random.default_rng(1)
b=bases.copy(); pos=rng.integers(0,b.size,b.size//200); b[pos]=np.frombuffer(b"NnacgtRY",dtype=np.uint8)[rng.integers(0,8,pos.size)]
seqs=synth.split_reads(b,offsets)
sc=hiplib.HipScanner(0); sc.set_patterns(pats); sc.upload(0,b,offsets)
prm=hiplib.make_params(min_len=1000,min_count=-1,flags=1|2|4|8)
sc.scan(0,prm); sc.sync(); res=sc.results(0).copy(); s1,wo=sc.window_sums(0)
prm.flags|=16
sc.scan(0,prm); sc.sync(); s2,_=sc.window_sums(0); raw,_=sc.window_raw(0)
for rep in range(5):
    prm.flags=1|2|4|8
    sc.scan(0,prm); sc.sync(); t1,_=sc.window_sums(0)
    prm.flags|=16
    sc.scan(0,prm); sc.sync(); t2,_=sc.window_sums(0); rw,_=sc.window_raw(0)
    print('rep',rep,'s1!=t1',int((s1!=t1).sum()),'t1!=t2',int((t1!=t2).sum()),'raw!=t2',int((rw.astype(np.int64).sum(axis=1)!=t2).sum()))
bad=np.nonzero(s1!=s2)[0]
print('mismatch', len(bad))
reads=np.searchsorted(wo,bad,side='right')-1
for bi,ri in list(zip(bad,reads))[:12]:
    w=bi-wo[ri]
    tail=["forward","reverse"][res['tail'][ri]]
    _,counts=orc.window_count_matrix(seqs[ri],tail,pats,100,6,100,20000)
    print('read',ri,'win',w,'tilewin',w%487,'lane',(w%487)//8,'j',(w%487)%8,'sums-only',s1[bi],'raw-run',s2[bi],'oracle',counts[w].sum(), 'has N nearby', 'N' in seqs[ri].upper()[max(0,100+6*w-50):100+6*w+150] if tail=='forward' else '?')
