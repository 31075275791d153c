#!/usr/bin/env python3
"""Condense a scripts/profile.sh output directory (gpurun_out/prof_<tag>) into profiles/<name>/:
the rocprofv3 --stats kernel summary as-is plus one CSV of per-launch mean PMC counters."""
import collections
import csv
import glob
import os
import shutil
import sys

src, name = sys.argv[1], sys.argv[2]
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", name)
os.makedirs(dst, exist_ok=True)
for f in glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")):
    shutil.copyfile(f, os.path.join(dst, "kernel_stats.csv"))
# the default command (two contexts issuing consecutive batches): the launches overlap and stretch each other
for f in glob.glob(os.path.join(src, "trace_default", "*", "*_kernel_stats.csv")):
    shutil.copyfile(f, os.path.join(dst, "kernel_stats_default_command_two_streams.csv"))
rows = []
for f in sorted(glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv"))):
    acc = collections.defaultdict(list)
    meta = {}
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("tps_"):
            acc[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
            meta[r["Kernel_Name"]] = (r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"], r["SGPR_Count"], r["Scratch_Size"])
    for (k, c), v in sorted(acc.items()):
        rows.append([k, c, len(v), sum(v) / len(v)] + list(meta[k]))
with open(os.path.join(dst, "pmc_per_launch_mean.csv"), "w", newline="") as h:
    w = csv.writer(h)
    w.writerow(["kernel", "counter", "launches", "mean_value", "grid", "workgroup", "lds_block", "vgpr", "sgpr", "scratch"])
    w.writerows(rows)
print(open(os.path.join(dst, "kernel_stats.csv")).read())
for r in rows:
    print(r[:4])
