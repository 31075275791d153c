#!/usr/bin/env python3
"""Kernel time of one scan configuration outside bench.py's workloads: scripts/time_scan.py MOTIF K SLIDE FLAGS [N_READS [READ_LEN]]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from topsicle_amd import hiplib, synth, allsteps
motif, k, slide, flags = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
n = int(sys.argv[5]) if len(sys.argv) > 5 else 10000
rl = int(sys.argv[6]) if len(sys.argv) > 6 else 25000
b, o, _ = synth.make_reads(n, rl, motif, 20250920, errors=synth.ONT)
sc = hiplib.HipScanner(0)
sc.set_patterns(allsteps.patterns_to_search(motif, k))
for s in range(4):
    sc.upload(s, b, o)
prm = hiplib.make_params(min_len=9000, min_count=allsteps.min_count_for_cutoff(0.7, 1000 / len(motif), 1000), slide=slide, flags=flags)
for i in range(40):
    sc.scan(i % 4, prm)
sc.sync()
sc.kernel_time_reset()
for i in range(200):
    sc.scan(i % 4, prm)
sc.sync()
nl, tot, mean = sc.kernel_time_ms()
print(sc.kernel_info(0), "kernel %.4f ms over %d launches" % (mean, nl))
