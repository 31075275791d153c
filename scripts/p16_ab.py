#!/usr/bin/env python3
"""GPU box: an 8-letter motif at the reference's default k = len - 2 (TTTTAGGG, k = 6: sixteen patterns, slide 8) -- kernel time with the
fused tiles (round 5) against the generic kernel that such tables took before (debug option force_generic), and parity of both against
the C oracle on every read of a smaller batch.   usage: p16_ab.py [n_reads]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402

import oracle_c  # noqa: E402
import topsicle_oracle as orc  # noqa: E402
from topsicle_amd import allsteps, hiplib, synth  # noqa: E402

motif, k, slide = "TTTTAGGG", 6, 8
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
pats = allsteps.patterns_to_search(motif, k)
assert len(pats) == 16
prm = hiplib.make_params(min_len=9000, min_count=allsteps.min_count_for_cutoff(0.7, 1000 / len(motif), 1000), slide=slide,
                         flags=hiplib.F_STEP1 | hiplib.F_WINDOWS | hiplib.F_BINSEG | hiplib.F_STORE_SUMS)
bases, offsets, _ = synth.make_reads(n, 15000, motif, seed=20250919 + 16, errors=synth.ONT)
for name, opts in (("fused", {}), ("generic", {"force_generic": 1})):
    sc = hiplib.HipScanner(0)
    for key, val in opts.items():
        sc.debug_option(key, val)
    sc.set_patterns(pats)
    sc.upload(0, bases, offsets)
    for _ in range(20):
        sc.scan(0, prm)
    sc.sync()
    sc.kernel_time_reset()
    for _ in range(200):
        sc.scan(0, prm)
    sc.sync()
    nl, tot, mean = sc.kernel_time_ms()
    res = sc.results(0)
    m = 600
    # every window sum of the first m reads (per-read checksums), pass / tail / change point: against oracle.c
    out, ck = oracle_c.batch_ck(bases[: offsets[m]], offsets[: m + 1], pats, len(motif), 1000, 9000, 0.7, 100, slide, 100, 20000, threads=8)
    sums, win_off = sc.window_sums(0)
    got_ck = oracle_c.checksums(sums, win_off[: m + 1])
    bad = 0
    for i in range(m):
        if bool(out[i, 0]) != bool(res["pass"][i]):
            bad += 1
        elif out[i, 0] and (int(out[i, 5]) != int(res["bkp"][i]) or int(out[i, 1]) != int(res["tail"][i]) or int(ck[i, 0]) != int(got_ck[i])):
            bad += 1
    print(f"{name:8s} {sc.kernel_info(0)}  kernel {mean * 1e3:8.1f} us over {nl} launches; {int(res['pass'].sum())} of {n} reads pass; "
          f"mismatches vs oracle.c on {m} reads: {bad}")
    sc.close()
