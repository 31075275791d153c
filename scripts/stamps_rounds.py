"""Diagnostic: timeline of the scan kernel's phases (lane-0 cycle stamps per read), first round of wave slots vs later reads."""
import ctypes as C, os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from topsicle_amd import hiplib, synth, allsteps
motif, k, slide = "CCCTAA", 4, 6
pats = allsteps.patterns_to_search(motif, k)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
b, o, _ = synth.make_reads(n, 15000, motif, 1)
sc = hiplib.HipScanner(0); sc.set_patterns(pats)
sc.upload(0, b, o)
prm = hiplib.make_params(min_len=9000, min_count=100, slide=slide, flags=1 | 2 | 4 | 8)
for _ in range(300):
    sc.scan(0, prm)
sc.sync()
sc.debug_option("stamps", 1)
sc.scan(0, prm); sc.sync()
st = np.zeros((n, 16), np.uint64)
sc.lib.tps_debug_stamps_get.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
assert sc.lib.tps_debug_stamps_get(sc._h, 0, st.ctypes.data_as(C.c_void_p), n) == 0
st = st.astype(np.int64)
t0 = st[:, 0].min()
names = {0: "start", 1: "misc zeroed", 2: "heads staged", 3: "trc counted", 4: "decided", 5: "tile0 staged", 6: "t0 ph1", 7: "t0 xt", 11: "t0 ph2",
         12: "t0 rowscan", 8: "t0 done", 9: "tiles done", 10: "result"}
order = [0, 1, 2, 3, 4, 5, 6, 7, 11, 12, 8, 9, 10]
first = st[:, 0] - t0 < (st[:, 10].max() - t0) * 0.2
print("reads in the first round:", int(first.sum()), "later:", int((~first).sum()), "kernel span (clocks):", st[:, 10].max() - t0)
for label, sel in (("first round", first), ("later reads", ~first)):
    if not sel.any():
        continue
    s = st[sel]
    print(label, " start (since kernel start): mean %.0f  min %.0f  max %.0f" % ((s[:, 0] - t0).mean(), (s[:, 0] - t0).min(), (s[:, 0] - t0).max()))
    prev = 0
    for i in order[1:]:
        if (s[:, i] == 0).all():
            continue
        d = s[:, i] - s[:, prev]
        print("  %-14s +%8.0f clocks (median %8.0f)   at %8.0f since the read's start" % (names[i], d.mean(), np.median(d), (s[:, i] - s[:, 0]).mean()))
        prev = i
