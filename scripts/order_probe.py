"""Diagnostic: how much of a lone launch is lost to the ORDER in which reads meet wave slots.
Takes every read's duration from the clock stamps of the diagnostics build, then times the product kernel on the same
reads in other orders: as generated, longest first (the list scheduler's best case), shortest first, and grouped (the
reads of one workgroup have similar durations, groups in random order).  An upper bound for what dynamic claiming of
reads by waves, or any sorting by predicted work, could buy.
usage: python3 scripts/order_probe.py [K [FLAGS [N_READS [READ_LEN]]]]   (needs topsicle_amd/libtopsicle_hip_diag.so)"""
import ctypes as C, os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT)
import numpy as np
from topsicle_amd import hiplib, synth, allsteps

motif, slide = "CCCTAA", 6
k = int(sys.argv[1]) if len(sys.argv) > 1 else 6
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 15
n = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
rl = int(sys.argv[4]) if len(sys.argv) > 4 else 25000
pats = allsteps.patterns_to_search(motif, k)
RAGGED = os.environ.get("TPS_PROBE_RAGGED", "0") != "0"
if RAGGED:                       # log-normal lengths (median 11 kb, up to 60 kb): what a real ONT file looks like
    b, o, _ = synth.make_ragged_reads(n, motif, 20250920)
else:
    b, o, _ = synth.make_reads(n, rl, motif, 20250920)
lens = np.diff(o)
prm = hiplib.make_params(min_len=9000, min_count=allsteps.min_count_for_cutoff(0.7, 1000 / len(motif), 1000), slide=slide, flags=flags)


def durations():
    sc = hiplib.HipScanner(0, lib_path=os.path.join(ROOT, "topsicle_amd", "libtopsicle_hip_diag.so"))
    sc.set_patterns(pats)
    sc.upload(0, b, o)
    for _ in range(50):
        sc.scan(0, prm)
    sc.sync()
    sc.debug_option("stamps", 1)
    sc.scan(0, prm); sc.sync()
    st = sc.stamps(0).astype(np.int64)
    info = sc.kernel_info(0)
    sc.close()
    return (st[:, 10] - st[:, 13]).astype(np.float64) / 2400.0, info      # us at 2.4 GHz


def timed(order, label, sc):
    oo = np.zeros(n + 1, np.int64)
    np.cumsum(lens[order], out=oo[1:])
    bb = np.concatenate([b[o[i]:o[i + 1]] for i in order]) if RAGGED else b.reshape(n, rl)[order].reshape(-1)
    sc.upload(0, bb, oo)
    for _ in range(20):
        sc.scan(0, prm)
    sc.sync()
    best = []
    for _ in range(5):
        sc.kernel_time_reset()
        for _ in range(200):
            sc.scan(0, prm)
            sc.sync()                      # launches alone, like the serialised profile
        cnt, ms, _x = sc.kernel_time_ms()
        best.append(ms / max(cnt, 1) * 1e3)
    print("%-44s %8.2f us per launch (median of 5 x 200; min %.2f max %.2f)" % (label, float(np.median(best)), min(best), max(best)), flush=True)


dur, info = durations()
print(info)
print("read durations: mean %.2f us  median %.2f  p90 %.2f  max %.2f" % (dur.mean(), np.median(dur), np.percentile(dur, 90), dur.max()))
rng = np.random.default_rng(1)
sc = hiplib.HipScanner(0)
sc.set_patterns(pats)
sc.debug_option("event_stride", 1)
ident = np.arange(n)
timed(ident, "as generated", sc)
timed(rng.permutation(n), "random permutation", sc)
desc = np.argsort(-dur, kind="stable")
timed(desc, "longest first", sc)
timed(desc[::-1], "shortest first", sc)
wpg = 8
g = desc[: n - n % wpg].reshape(-1, wpg)
g = g[rng.permutation(len(g))].reshape(-1)
timed(np.concatenate([g, desc[n - n % wpg:]]), "similar durations per workgroup, random groups", sc)
if RAGGED:                       # what the host knows before the scan: the bases a read will have scanned (none if the length filter drops it)
    work = np.where(lens > 9000, np.minimum(lens, 20000), 0)
    timed(np.argsort(-work, kind="stable"), "by scanned length, longest first (known a priori)", sc)
    timed(np.argsort(-(work // 2048), kind="stable"), "... in classes of 2048 bases", sc)
# the longest fifth first, the rest as generated: what a coarse predictor would give
cut = np.percentile(dur, 80)
timed(np.concatenate([ident[dur >= cut], ident[dur < cut]]), "longest fifth first, rest as generated", sc)
