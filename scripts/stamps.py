"""Diagnostic: per-phase clock shares of the scan kernel (thread-0 stamps per read)."""
import ctypes as C, os, sys
# needs the diagnostics build: python -c "import __graft_entry__ as g; g.build_hip_diag()"
os.environ.setdefault("TOPSICLE_HIP_LIB", os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "topsicle_amd", "libtopsicle_hip_diag.so"))
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from topsicle_amd import hiplib, synth, allsteps
motif, k, slide = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else ("CCCTAA", 4, 6)
pats = allsteps.patterns_to_search(motif, k)
b, o, _ = synth.make_reads(10000, 15000, motif, 1)
sc = hiplib.HipScanner(0); sc.set_patterns(pats)
sc.upload(0, b, o)
prm = hiplib.make_params(min_len=9000, min_count=100, slide=slide, flags=1 | 2 | 4 | 8)
sc.scan(0, prm); sc.sync()
sc.debug_option("stamps", 1)
sc.scan(0, prm); sc.sync()
st = np.zeros((10000, 16), np.uint64)
sc.lib.tps_debug_stamps_get.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
rc = sc.lib.tps_debug_stamps_get(sc._h, 0, st.ctypes.data_as(C.c_void_p), 10000)
assert rc == 0
st = st.astype(np.int64)
tot = (st[:, 10] - st[:, 0])
print("mean total clocks per read:", tot.mean(), " kernel span clocks:", st[:, 10].max() - st[:, 0].min())
segs = [("misc zero", 0, 1), ("stage heads", 1, 2), ("trc count", 2, 3), ("trc sum+decide", 3, 4), ("stage tile0", 4, 5),
        ("t0 blocks (ph1)", 5, 6), ("t0 XT scan", 6, 7), ("t0 windows (ph2)", 7, 11), ("t0 row scan", 11, 12), ("t0 store+cand (ph3)", 12, 8),
        ("rest tiles", 8, 9), ("binseg+result", 9, 10)]
for n, a_, b_ in segs:
    dd = (st[:, b_] - st[:, a_])
    if (st[:, b_] == 0).all() or (st[:, a_] == 0).all():
        continue
    print(f"{n:20s} {dd.mean():10.0f}  {100 * dd.mean() / tot.mean():5.1f}%")
