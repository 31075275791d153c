"""Where the end-to-end time goes (GPU box): cProfile of batch.EnginePool.scan_file and of the CLI on a FASTQ of config-2 reads."""
import cProfile, io, os, pstats, shutil, sys, tempfile, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from topsicle_amd import allsteps, batch, e2e, hiplib, main as cli, seqio, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
b, o, _ = synth.make_reads(n, 15000, "CCCTAA", seed=20250920)
tmp = tempfile.mkdtemp(prefix="tps_prof_")
fq = os.path.join(tmp, "reads.fastq")
e2e.write_fastq(fq, b, o)
pats = allsteps.patterns_to_search("CCCTAA", 4)
engines = [hiplib.HipScanner(0) for _ in range(2)]
ep = batch.EnginePool(engines, pats)
prm = hiplib.make_params(min_len=9000, min_count=allsteps.min_count_for_cutoff(0.7, 1000 / 6, 1000))
for rep in range(3):
    t0 = time.perf_counter()
    stamps = []
    for pb, res, *_ in ep.scan_file(fq, prm):
        stamps.append(round((time.perf_counter() - t0) * 1e3, 1))
    print("scan_file pass", rep, "batch arrival ms:", stamps, "total", round((time.perf_counter() - t0) * 1e3, 1))
pr = cProfile.Profile()
pr.enable()
for pb, res, *_ in ep.scan_file(fq, prm):
    pass
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18); print(s.getvalue()[:3500])
for e in engines: e.close()
od = os.path.join(tmp, "out")
argv = ["--inputDir", fq, "--outputDir", od, "--pattern", "CCCTAA", "--telophrase", "4", "--slide", "6"]
e2e._quiet(cli.main, argv)
cli.wait_for_plots()
shutil.rmtree(od)
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
e2e._quiet(cli.main, argv)
pr.disable()
print("cli total ms", round((time.perf_counter() - t0) * 1e3, 1), cli.LAST_TIMINGS)
cli.wait_for_plots()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(30); print(s.getvalue()[:6000])
shutil.rmtree(tmp)
