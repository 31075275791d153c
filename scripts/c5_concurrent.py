"""Experiment: the three table passes of BASELINE configs[4] on THREE contexts (streams) at once instead of back to back on one."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from topsicle_amd import hiplib, synth, allsteps
motif = "CCCTAA"
b, o, _ = synth.make_reads(10000, 25000, motif, seed=20250919 + 4, errors=synth.ONT)
prm = hiplib.make_params(no_bp=1000, min_len=9000, min_count=allsteps.min_count_for_cutoff(0.7, 1000 / 6, 1000), window=100, slide=6,
                         trimfirst=100, maxlen=20000, flags=31)
tables = [allsteps.patterns_to_search(motif, k) for k in (4, 5, 6)]
copies = 3
scs = [hiplib.HipScanner(0) for _ in tables]
for sc, t in zip(scs, tables):
    sc.set_patterns(t)
    for s in range(copies):
        sc.upload(s, b, o)
def step_conc(i):
    for sc in scs:
        sc.scan(i % copies, prm)
def sync():
    for sc in scs:
        sc.sync()
for i in range(60):
    step_conc(i)
sync()
n = 200
t0 = time.perf_counter()
for i in range(n):
    step_conc(i)
sync()
dt = time.perf_counter() - t0
print(f"three contexts at once: {dt / n * 1e6:.1f} us per step")
# back to back on ONE context
one = scs[0]
def step_seq(i):
    for t in tables:
        one.set_patterns(t)
        one.scan(i % copies, prm)
for i in range(30):
    step_seq(i)
one.sync()
t0 = time.perf_counter()
for i in range(n):
    step_seq(i)
one.sync()
dt = time.perf_counter() - t0
print(f"one context, back to back: {dt / n * 1e6:.1f} us per step")
