"""Diagnostic: distribution of per-read durations (clock stamps 0 and 10 of every read) -- stragglers set the
makespan of a launch.  Needs the diagnostics build (build_hip_diag).  usage: stamps_dist.py MOTIF K SLIDE [N_READS READ_LEN RAW]"""
import ctypes as C, os, sys
os.environ.setdefault("TOPSICLE_HIP_LIB", os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "topsicle_amd", "libtopsicle_hip_diag.so"))
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from topsicle_amd import hiplib, synth, allsteps
motif, k, slide = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 10000
rl = int(sys.argv[5]) if len(sys.argv) > 5 else 25000
raw = int(sys.argv[6]) if len(sys.argv) > 6 else 0
pats = allsteps.patterns_to_search(motif, k)
b, o, _ = synth.make_reads(n, rl, motif, 20250919 + 4)
sc = hiplib.HipScanner(0); sc.set_patterns(pats)
sc.upload(0, b, o)
prm = hiplib.make_params(min_len=9000, min_count=allsteps.min_count_for_cutoff(0.7, 1000 / len(motif), 1000), slide=slide, flags=1 | 2 | 4 | 8 | (16 if raw else 0))
sc.scan(0, prm); sc.sync()
sc.debug_option("stamps", 1)
sc.kernel_time_reset()
sc.scan(0, prm); sc.sync()
print("kernel", sc.kernel_info(0), "ms", sc.kernel_time_ms()[2])
st = np.zeros((n, 16), np.uint64)
sc.lib.tps_debug_stamps_get.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
assert sc.lib.tps_debug_stamps_get(sc._h, 0, st.ctypes.data_as(C.c_void_p), n) == 0
st = st.astype(np.int64)
res = sc.results(0)
ok = (st[:, 0] > 0) & (st[:, 10] > 0)
print("reads with stamps:", int(ok.sum()))
st = st[ok]; res = res[ok]
tot = st[:, 10] - st[:, 0]
span = st[:, 10].max() - st[:, 0].min()
print("reads", n, "pass", int(res["pass"].sum()), "span clocks", span)
for q in (50, 90, 99, 99.9, 100):
    print(f"  p{q}: {np.percentile(tot, q):.0f} clocks")
order = np.argsort(-tot)[:5]
print("slowest reads:", [(int(i), int(tot[i]), int(res["n_win"][i])) for i in order])
# concurrency over time: how many reads are alive at 20 sample points
t0 = st[:, 0].min()
for f in np.linspace(0.05, 0.95, 10):
    t = t0 + f * span
    print(f"  t={f:.2f}: alive {int(((st[:, 0] <= t) & (st[:, 10] > t)).sum())}")
