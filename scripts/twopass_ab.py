"""A/B on one box: file -> results on the config-2 file (every read telomeric: the case in which auto mode must not cost anything)
and on a 1 %-telomeric 30 kb file, two_pass off / auto / on, runs interleaved.  usage: python scripts/twopass_ab.py"""
import os, sys, time, tempfile, shutil, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from topsicle_amd import allsteps, batch, e2e, hiplib, synth

def run(fq, n_bases, modes, reps=7):
    pats = allsteps.patterns_to_search("CCCTAA", 4)
    prm = hiplib.make_params(no_bp=1000, min_len=9000, min_count=allsteps.min_count_for_cutoff(0.7, 1000 / 6, 1000), window=100, slide=6, trimfirst=100, maxlen=20000)
    engines = [hiplib.HipScanner(0), hiplib.HipScanner(0)]
    pools = {m: batch.EnginePool(engines, pats, two_pass=m) for m in modes}
    times = {m: [] for m in modes}
    for r in range(reps + 1):
        for m in modes:
            t0 = time.perf_counter()
            n = sum(pb.n for pb, *_ in pools[m].scan_file(fq, prm))
            times[m].append(time.perf_counter() - t0)
    out = {m: dict(median_ms=round(1e3 * float(np.median(times[m][1:])), 3), best_ms=round(1e3 * min(times[m][1:]), 3),
                   bases_per_s=n_bases / float(np.median(times[m][1:])), heads_batches=pools[m].stats["heads_batches"], batches=pools[m].stats["batches"]) for m in modes}
    for e in engines:
        e.close()
    return out

tmp = tempfile.mkdtemp(prefix="tps_ab_")
try:
    b, o, _ = synth.make_reads(10000, 15000, "CCCTAA", seed=20250920)
    fq = os.path.join(tmp, "c2.fastq"); e2e.write_fastq(fq, b, o)
    print("config2 file (all telomeric)", json.dumps(run(fq, int(o[-1]), ["off", "auto", "on"])))
    b, o, _ = synth.make_reads(10000, 30000, "CCCTAA", seed=20250921, telomeric_fraction=0.01)
    fq = os.path.join(tmp, "wgs.fastq"); e2e.write_fastq(fq, b, o)
    print("30 kb reads, 1 % telomeric", json.dumps(run(fq, int(o[-1]), ["off", "auto", "on"])))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
