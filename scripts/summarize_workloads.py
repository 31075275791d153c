#!/usr/bin/env python3
"""Condense scripts/profile_workloads.sh output (gpurun_out/profw_<tag>) into one table per workload:
kernel, calls, mean ns (rocprofv3 --stats), FETCH_SIZE / WRITE_SIZE per launch (KB; FETCH doubled = HBM bytes on gfx950),
instruction counters, and the bench line's algorithmic bytes -> fractions of the 8 TB/s roofline.
With a second argument the table and the per-workload kernel_stats.csv are copied to profiles/<name>/."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src = sys.argv[1]
dst = None
if len(sys.argv) > 2:
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", sys.argv[2])
    os.makedirs(dst, exist_ok=True)
rows = []
for d in sorted(glob.glob(os.path.join(src, "*_f*"))):
    name = os.path.basename(d)
    stats = glob.glob(os.path.join(d, "trace", "*", "*_kernel_stats.csv"))
    kern = calls = mean_ns = None
    if stats:
        ks = [(r["Name"], int(r["Calls"]), float(r["AverageNs"])) for r in csv.DictReader(open(stats[0])) if r["Name"].startswith("tps_scan")]
        ks = [k for k in ks if k[1] * 4 >= max(x[1] for x in ks)]          # (drop stray launches of the set-up phase)
        if len(ks) == 1:
            kern, calls, mean_ns = ks[0]
        elif ks:
            # several scan kernels per step (the three k passes of config5): one step = one launch of each, so the step's
            # kernel time is the sum of the means weighted by launches per step
            per_step = min(k[1] for k in ks)
            kern = " + ".join(f"{k[0]}x{round(k[1] / per_step)}" for k in ks)
            calls = per_step
            mean_ns = sum(k[1] * k[2] for k in ks) / per_step
        if dst:
            shutil.copyfile(stats[0], os.path.join(dst, f"{name}_kernel_stats.csv"))
    pmc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "pmc_*", "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("tps_"):
                pmc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    pm = {k: sum(v) / len(v) for k, v in pmc.items()}
    bench = None
    try:
        for ln in open(os.path.join(d, "bench.json")):
            if ln.startswith("{"):
                bench = json.loads(ln)
    except OSError:
        pass
    alg = bench["roofline"]["algorithmic_bytes_per_launch"] if bench else None
    live = bench["roofline"]["kernel_ms_mean"] if bench else None
    step_us = round(bench["ms_per_step"] * 1e3, 2) if bench else None
    # table passes that OVERLAP on the GPU (config 5: one stream per table): the kernels' own durations overlap too, what bounds
    # them together is the step -- the wall time per step of the traced run stands in for the launch duration
    overlapping = bool(bench) and "overlapping" in str(bench["roofline"].get("duration_source", ""))
    if overlapping and mean_ns is not None:
        kern = (kern or "") + " (overlapping: duration = traced step)"
    dur_ns = bench["ms_per_step"] * 1e6 if overlapping else mean_ns
    traffic = (2 * pm["FETCH_SIZE"] + pm["WRITE_SIZE"]) * 1024 if "FETCH_SIZE" in pm and "WRITE_SIZE" in pm else None
    row = dict(workload=name, kernel=kern, calls=calls, rocprof_mean_us=None if mean_ns is None else round(mean_ns / 1e3, 2),
               live_event_mean_us=None if live is None else round(live * 1e3, 2),
               step_us=step_us,
               algorithmic_MB=None if alg is None else round(alg / 1e6, 2),
               frac_of_8TBs=None if not (alg and dur_ns) else round(alg / (dur_ns * 1e-9) / 8e12, 4),
               hbm_traffic_MB=None if traffic is None else round(traffic / 1e6, 2),
               traffic_frac=None if not (traffic and dur_ns) else round(traffic / (dur_ns * 1e-9) / 8e12, 4))
    for c in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
        row[c] = round(pm[c]) if c in pm else None
    rows.append(row)
if rows:
    w = csv.DictWriter(sys.stdout, fieldnames=list(rows[0]))
    w.writeheader()
    w.writerows(rows)
    if dst:
        with open(os.path.join(dst, "workloads.csv"), "w", newline="") as h:
            w = csv.DictWriter(h, fieldnames=list(rows[0]))
            w.writeheader()
            w.writerows(rows)
