"""Diagnostic: occupancy timeline of one scan launch from the per-read clock stamps (diagnostics build: -DTPS_STAMPS).
The shader clock counters of the 8 XCDs are not synchronised, so every read is placed on its own XCD's time axis
(workgroup w runs on XCD w % 8; the axis starts at the XCD's first stamp).  Prints, per XCD-averaged time slice, the number
of reads in flight, and the duration of reads by start time -- where a launch's time goes: ramp, full rounds, tail.
usage: TOPSICLE_HIP_LIB=.../libtopsicle_hip_diag.so python3 scripts/stamps_timeline.py [N_READS [READ_LEN [WPG]]]"""
import ctypes as C, os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from topsicle_amd import hiplib, synth, allsteps
motif, k, slide = "CCCTAA", int(os.environ.get("TPS_STAMP_K", "4")), 6
pats = allsteps.patterns_to_search(motif, k)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
rl = int(sys.argv[2]) if len(sys.argv) > 2 else 15000
wpg = int(sys.argv[3]) if len(sys.argv) > 3 else 4
b, o, _ = synth.make_reads(n, rl, motif, 20250920, errors={"ont": synth.ONT, "hifi": synth.HIFI}[os.environ.get("TPS_STAMP_ERRORS", "ont")], telomeric_fraction=float(os.environ.get("TPS_TELO_FRAC", "1")))
sc = hiplib.HipScanner(0); sc.set_patterns(pats)
sc.upload(0, b, o)
prm = hiplib.make_params(min_len=9000, min_count=allsteps.min_count_for_cutoff(0.7, 1000 / len(motif), 1000), slide=slide, flags=int(os.environ.get("TPS_STAMP_FLAGS", "15")))
for _ in range(300):
    sc.scan(0, prm)
sc.sync()
sc.debug_option("stamps", 1)
sc.scan(0, prm); sc.sync()
st = np.zeros((n, 16), np.uint64)
sc.lib.tps_debug_stamps_get.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
assert sc.lib.tps_debug_stamps_get(sc._h, 0, st.ctypes.data_as(C.c_void_p), n) == 0
print(sc.kernel_info(0))
st = st.astype(np.int64)
# device-wide timeline from the 100 MHz real-time counter stamped at kernel entry (15); a read's end = its entry + its duration in
# shader clocks (13 -> 10) converted with the clock ratio measured over the whole launch
rt = st[:, 15].astype(np.float64)
rt -= rt.min()
rt *= 10.0                                                  # ns
cyc = (st[:, 10] - st[:, 13]).astype(np.float64)            # shader clocks from kernel entry to the result
ghz = float(os.environ.get("TPS_CLK_GHZ", "2.4"))
start = rt
end = rt + cyc / ghz
dur = end - start
span = end.max()
print(f"reads {n}  span {span / 1e3:.1f} us   entry->table barrier {np.mean(st[:, 14] - st[:, 13]):.0f} clocks (median {np.median(st[:, 14] - st[:, 13]):.0f}, p90 {np.percentile(st[:, 14] - st[:, 13], 90):.0f})   barrier->scan start {np.mean(st[:, 0] - st[:, 14]):.0f}")
print(f"read duration incl. prologue: mean {dur.mean() / 1e3:.2f} us  median {np.median(dur) / 1e3:.2f}  p90 {np.percentile(dur, 90) / 1e3:.2f}  max {dur.max() / 1e3:.2f};  inside scan_read {np.mean(st[:, 10] - st[:, 0]):.0f} clocks")
nb = 28
edges = np.linspace(0, span, nb + 1)
print("slice_start_us  in_flight  started  finished  mean_dur_us_of_reads_started_here")
for i in range(nb):
    lo, hi = edges[i], edges[i + 1]
    mid = (lo + hi) / 2
    inflight = ((start <= mid) & (end > mid)).sum()
    s = (start >= lo) & (start < hi)
    f = (end >= lo) & (end < hi)
    md = dur[s].mean() / 1e3 if s.any() else 0
    print(f"{lo / 1e3:10.2f}  {inflight:8d}  {int(s.sum()):8d}  {int(f.sum()):8d}  {md:10.2f}")
names = {14: "table barrier", 0: "scan_read entry", 1: "misc zeroed", 2: "heads staged", 3: "trc counted", 4: "decided", 5: "tile0 staged", 6: "t0 ph1", 7: "t0 xt / pp windows", 11: "t0 ph2 / pp rows out",
         12: "t0 rowscan/ph2 / pp candidates", 8: "t0 done", 9: "tiles done", 10: "result"}
order = [13, 14, 0, 1, 2, 3, 4, 5, 6, 7, 11, 12, 8, 9, 10]
early = start < np.percentile(start, 30)
for label, sel in (("reads started early (first 30 %)", early), ("reads started late", ~early)):
    s = st[sel]
    print(label, int(sel.sum()))
    prev = 13
    for i in order[1:]:
        ok = (s[:, i] != 0) & (s[:, prev] != 0)      # (reads that fail the TRC filter never stamp the tile phases)
        if not ok.any():
            continue
        d = (s[:, i] - s[:, prev])[ok]
        print("  %-16s +%8.0f clocks (median %8.0f)  [%d reads]" % (names[i], d.mean(), np.median(d), int(ok.sum())))
        if ok.sum() * 2 > len(s):
            prev = i
