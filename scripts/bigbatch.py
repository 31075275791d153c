"""Stress: one resident batch larger than 4 GiB of bases (64-bit offsets / window offsets everywhere).
A 20k-read block is tiled, so read i and read i mod 20000 must give identical results and window sums."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from topsicle_amd import hiplib, synth, allsteps
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
pats = allsteps.patterns_to_search("CCCTAA", 4)
b, o, _ = synth.make_reads(20000, 8000, "CCCTAA", 3, tract_min=500, tract_max=4000)
n0 = 20000
bases = np.tile(b, reps)
lens = np.tile(np.diff(o), reps)
offsets = np.zeros(len(lens) + 1, np.int64); np.cumsum(lens, out=offsets[1:])
print(f"batch: {len(lens)} reads, {bases.size / 2**30:.2f} GiB of bases")
sc = hiplib.HipScanner(0); sc.set_patterns(pats)
t0 = time.time(); sc.upload(0, bases, offsets); print(f"upload {time.time() - t0:.2f} s")
prm = hiplib.make_params(min_len=5000, min_count=allsteps.min_count_for_cutoff(0.7, 1000 / 6, 1000), flags=15)
sc.scan(0, prm); sc.sync(); sc.kernel_time_reset()
for _ in range(3): sc.scan(0, prm)
sc.sync(); _n, _t, mean = sc.kernel_time_ms(); print(f"scan kernel {mean:.3f} ms = {mean * 1e6 / len(lens):.2f} ns/read, {bases.size / mean / 1e6:.0f} G bases/s")
res = sc.results(0).copy()
sums, win_off = sc.window_sums(0)
print("window sums:", sums.size, "entries; last window offset", int(win_off[-1]))
ok = True
for f in ("best_start", "best_end", "tail", "pass", "n_win", "bkp"):
    a = res[f].reshape(reps, n0)
    ok &= bool((a == a[0]).all())
first = sums[: win_off[n0]]
for rep in (1, reps // 2, reps - 1):
    lo, hi = win_off[rep * n0], win_off[(rep + 1) * n0]
    ok &= bool(np.array_equal(sums[lo:hi], first))
print("periodic results:", ok, " passing reads:", int(res["pass"].sum()))
sys.exit(0 if ok else 1)
