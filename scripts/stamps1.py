"""Diagnostic: phase timeline of a read that has the CU to itself (n reads << machine)."""
import ctypes as C, os, sys
# needs the diagnostics build: python -c "import __graft_entry__ as g; g.build_hip_diag()"
os.environ.setdefault("TOPSICLE_HIP_LIB", os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "topsicle_amd", "libtopsicle_hip_diag.so"))
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from topsicle_amd import hiplib, synth, allsteps
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pats = allsteps.patterns_to_search("CCCTAA", 4)
b, o, _ = synth.make_reads(n, 15000, "CCCTAA", 1)
sc = hiplib.HipScanner(0); sc.set_patterns(pats)
sc.upload(0, b, o)
prm = hiplib.make_params(min_len=9000, min_count=100, slide=6, flags=1 | 2 | 4 | 8)
for _ in range(3):
    sc.scan(0, prm)
sc.sync()
sc.debug_option("stamps", 1)
sc.scan(0, prm); sc.sync()
st = np.zeros((n, 16), np.uint64)
sc.lib.tps_debug_stamps_get.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
assert sc.lib.tps_debug_stamps_get(sc._h, 0, st.ctypes.data_as(C.c_void_p), n) == 0
st = st.astype(np.int64)
tot = st[:, 10] - st[:, 0]
print(f"n={n}: mean clocks per read {tot.mean():.0f}; kernel span {st[:, 10].max() - st[:, 0].min()}")
segs = [("misc zero", 0, 1), ("stage heads", 1, 2), ("trc count", 2, 3), ("trc sum+decide", 3, 4), ("stage tile0", 4, 5),
        ("t0 blocks (ph1)", 5, 6), ("t0 XT scan", 6, 7), ("t0 FO+windows (ph2)", 7, 11), ("t0 row scan", 11, 12), ("t0 cand (ph3)", 12, 8),
        ("rest tiles", 8, 9), ("binseg+result", 9, 10)]
for nm, a_, b_ in segs:
    dd = st[:, b_] - st[:, a_]
    print(f"{nm:22s} {dd.mean():9.0f}  {100 * dd.mean() / tot.mean():5.1f}%")
