#!/usr/bin/env python3
"""bench.py -- throughput of the telomere scan hot path on MI355X.

A "step" is ONE pass of the hot path (step-1 TRC + sliding-window k-mer counts + single-split
change-point) over ONE resident batch of synthetic reads, through the C ABI of
libtopsicle_hip.so.  Workload at N=1 = BASELINE.json configs[1]: 10k synthetic ONT-like reads
x 15 kb, --pattern CCCTAA (k=4, 12 patterns), window=100 slide=6.  Several copies of the batch
are kept resident at distinct HBM addresses and scanned in turn so that no step is served from
the 256 MiB Infinity Cache (the batch alone is only 150 MB).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU, each scanning its own batch (weak scaling, no data-path collective);
torch.distributed (gloo) is used only for the barrier and the max-over-ranks time.  Run WITHOUT a launcher
(`python bench.py --gpus 8`, WORLD_SIZE unset) the script spawns its N ranks itself -- fresh child processes, started
before this process has touched a GPU -- and relays rank 0's JSON line; it fails if fewer than N GPUs are visible
(TPS_BENCH_SHARE_GPU=1 lets ranks share devices: the launcher test on a 1-GPU box).

Two timed regions of K steps each, both bracketed by barrier + device sync (round 4), each REPEATED R times inside the run (round 5:
R is chosen after the first repeat so that at least 50 ms of GPU work are timed whatever K is -- the driver's `--steps 20` is 0.9 ms of
it -- and at least 3; `ms_per_step` is the MEDIAN over the repeats, p10 / p90 beside it):
  1. strictly one launch after the other on one context -> the kernel's own duration (HIP events), `roofline`, `single_stream`,
     `value_single_stream`;
  2. the same K steps with consecutive batches on `--streams` (default 2) contexts, each with its own stream, as the file pipeline
     issues them (batch.EnginePool keeps two contexts per GPU) -> `value`, `ms_per_step`, `pipelined`.  A 10 000-read launch is
     1.63 rounds of the chip's 6 144 wave slots: alone, its last third runs on a draining GPU (the roofline says what that costs:
     frac 0.84 against steady_state_frac 0.97); with the next batch's launch already queued on the other stream its drain is the
     neighbour's ramp -- so `ms_per_step` is a two-launch-overlap figure and can lie BELOW the kernel's own duration
     (`roofline.kernel_ms_mean`); the line says so in `consistency`.  `--streams 1` skips region 2 (`value` = `value_single_stream`):
     the command for a rocprofv3 --stats summary whose average duration is that of an undisturbed launch (scripts/profile.sh adds
     `--no-steady`: the steady-state leg's 4x launches carry the same kernel name).
`roofline.steady_state` is measured in the run (40 launches of a batch four times the size, ~8 ms) unless `--no-steady`.
`roofline.traffic` and the `roofline.valu` object (instructions per read, lane-instructions per scanned base, VALU busy over one
launch and over the two-stream region) use the PMC counters of the newest committed profile of the same workload -- counters cannot
be collected inside a timed run -- and name their source file.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

# torch is only imported for the multi-process barrier and never touches a GPU here; main() dlopens the HIP library
# (hiplib.load_library(): no GPU call) BEFORE torch.distributed comes in, so that its libamdhip64 (ROCm 7.2, /opt/rocm)
# is the HIP runtime of the process.
from topsicle_amd import hiplib, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ~6290

CONFIGS = {
    # name: (n_reads, read_len, motif, k, window, slide, errors, seed index)
    "config2": dict(n_reads=10000, read_len=15000, motif="CCCTAA", k=4, window=100, slide=6,
                    errors=synth.ONT, seed=20250919 + 1,
                    desc="BASELINE configs[1]: 10k synthetic ONT reads x 15 kb, --pattern CCCTAA, window=100 slide=6"),
    "config4_sample": dict(n_reads=10000, read_len=30000, motif="CCCTAA", k=4, window=100, slide=6,
                           errors=synth.ONT, seed=20250919 + 3,
                           desc="BASELINE configs[3] sample: 10k synthetic ONT reads x 30 kb, --pattern CCCTAA (20 kb scanned per read)"),
    "config4_1pct": dict(n_reads=10000, read_len=30000, motif="CCCTAA", k=4, window=100, slide=6,
                         errors=synth.ONT, seed=20250919 + 3, telomeric_fraction=0.01,
                         desc="BASELINE configs[3] sample, 1 % of the reads telomeric (the step-1-dominated regime of real WGS data)"),
    "config4_01pct": dict(n_reads=10000, read_len=30000, motif="CCCTAA", k=4, window=100, slide=6,
                          errors=synth.ONT, seed=20250919 + 3, telomeric_fraction=0.001,
                          desc="BASELINE configs[3] sample, 0.1 % of the reads telomeric"),
    # BASELINE configs[4] runs one pass per k (--telophrase 4 5 6) with --rawcountpattern: a sample of one GPU's shard per k
    # (add --flags 31 for the raw counts; k = 5 and 6 use tables with self-overlapping k-mers)
    "config5_k4": dict(n_reads=10000, read_len=25000, motif="CCCTAA", k=4, window=100, slide=6, errors=synth.ONT, seed=20250919 + 4,
                       desc="BASELINE configs[4] sample, k=4 pass: 10k synthetic ONT reads x 25 kb, --pattern CCCTAA"),
    "config5_k5": dict(n_reads=10000, read_len=25000, motif="CCCTAA", k=5, window=100, slide=6, errors=synth.ONT, seed=20250919 + 4,
                       desc="BASELINE configs[4] sample, k=5 pass: 10k synthetic ONT reads x 25 kb, --pattern CCCTAA"),
    "config5_k6": dict(n_reads=10000, read_len=25000, motif="CCCTAA", k=6, window=100, slide=6, errors=synth.ONT, seed=20250919 + 4,
                       desc="BASELINE configs[4] sample, k=6 pass: 10k synthetic ONT reads x 25 kb, --pattern CCCTAA"),
    # all of BASELINE configs[4]'s k passes on one resident batch per step: upload once, three table scans with raw counts
    "config5": dict(n_reads=10000, read_len=25000, motif="CCCTAA", k=4, ks=[4, 5, 6], raw=True, window=100, slide=6, errors=synth.ONT, seed=20250919 + 4,
                    desc="BASELINE configs[4] sample: 10k synthetic ONT reads x 25 kb, --pattern CCCTAA --telophrase 4 5 6 --rawcountpattern "
                         "(one step = the three k passes over one resident batch)"),
    # not a BASELINE config: read lengths as a real ONT file has them (log-normal, median 11 kb, 60 b .. 60 kb; the reference's demo data span
    # 1.6 - 48 kb) -- the shape on which the order of the reads in a launch matters (tps::plan_dispatch_order; TOPSICLE_HIP_DEBUG=file_order=1 for the A/B)
    "ragged_ont": dict(n_reads=20000, read_len=0, ragged=True, motif="CCCTAA", k=4, window=100, slide=6, errors=synth.ONT, seed=20250919 + 7,
                       desc="diagnostic: 20k synthetic ONT reads of log-normal length (median 11 kb, up to 60 kb), --pattern CCCTAA"),
    "config3_per_gpu": dict(n_reads=25000, read_len=20000, motif="AAACCCT", k=5, window=100, slide=7,
                            errors=synth.HIFI, seed=20250919 + 2,
                            desc="BASELINE configs[2] shard: 25k synthetic HiFi reads x 20 kb per GPU, --pattern AAACCCT"),
}


def kmer_table(motif, k):
    """Reference-order pattern list (host logic of the product, topsicle_amd.allsteps)."""
    from topsicle_amd import allsteps
    return allsteps.patterns_to_search(motif, k)


def min_count_for_cutoff(cutoff, no_bp, motif_len):
    from topsicle_amd import allsteps
    return allsteps.min_count_for_cutoff(cutoff, no_bp / motif_len, no_bp)


def algorithmic_bytes(lens, passed, n_win, P, prm):
    """fixed formula, independent of implementation choices (1 B per scanned base + 4 B per
    window, + P bytes per window when the raw counts are stored)."""
    lens = np.asarray(lens, dtype=np.int64)
    step1 = (2 * np.minimum(lens, prm.no_bp) + 2 * P * 4).sum()
    n_s = np.maximum(np.minimum(lens, prm.maxlen) - prm.trimfirst, 0)
    step2 = (n_s[passed] + 4 * n_win[passed]).sum()
    if prm.flags & hiplib.F_STORE_RAW:
        step2 += (P * n_win[passed]).sum()
    step3 = (4 * n_win[passed] + 12).sum()
    return int(step1 + step2 + step3), int(step1), int(step2), int(step3)


def _profile_dirs():
    import glob
    import re

    def version(d):                      # profiles/r<round>_v<version>_<name>
        m = re.match(r"r(\d+)_v(\d+)", os.path.basename(d))
        return (int(m.group(1)), int(m.group(2))) if m else (-1, -1)
    return sorted(glob.glob(os.path.join(ROOT, "profiles", "r*")), key=version, reverse=True)


def profiled_traffic(workload, flags=0, ks=None):
    """HBM bytes per launch (per step for a several-table workload) of the scan kernel(s) and their per-launch counter means from
    the newest committed rocprofv3 PMC summary: config2 -> profiles/*/pmc_per_launch_mean.csv (FETCH_SIZE and WRITE_SIZE collected
    in separate --pmc passes of this same command; KB units; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950), the
    other workloads -> profiles/*/workloads.csv (scripts/profile_workloads.sh; a several-table step = its table passes' rows
    added up).  (None, None) when no committed profile covers the workload."""
    import csv
    PROFILED_COUNTERS.clear()
    for d in _profile_dirs():
        if workload == "config2" and not flags:
            f = os.path.join(d, "pmc_per_launch_mean.csv")
            if not os.path.exists(f):
                continue
            vals = {}
            for r in csv.DictReader(open(f)):
                if r["kernel"].startswith("tps_scan_kernel"):
                    vals[r["counter"]] = float(r["mean_value"])
            if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
                PROFILED_COUNTERS.update(vals)
                return int((2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024), os.path.relpath(f, ROOT)
            continue
        f = os.path.join(d, "workloads.csv")
        if not os.path.exists(f):
            continue
        rows = {r.get("workload", ""): r for r in csv.DictReader(open(f))}      # (older profile directories keep other columns)
        want = [f"{workload}_k{kk}_f31" for kk in ks] if (ks and len(ks) > 1) else [f"{workload}_f{flags}"]
        if not all(w in rows and rows[w].get("hbm_traffic_MB") and rows[w].get("SQ_INSTS_VALU") for w in want):
            continue
        try:
            for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES"):
                PROFILED_COUNTERS[c] = sum(float(rows[w][c]) for w in want)
            return int(sum(float(rows[w]["hbm_traffic_MB"]) for w in want) * 1e6), os.path.relpath(f, ROOT) + " (" + " + ".join(want) + ")"
        except (KeyError, ValueError):
            PROFILED_COUNTERS.clear()
    return None, None


PROFILED_COUNTERS = {}            # the scan kernel's per-launch counter means of that profile (SQ_INSTS_VALU, ...)
N_SIMDS = 256 * 4                 # MI355X: 256 CUs x 4 SIMDs; integer VALU issues one wave-instruction per 4 cycles per SIMD
ENGINE_CLOCK_HZ = 2.4e9           # MI355X_MICROARCH.md (peak engine clock)


def cpu_baseline(seqs, motif, k, prm, budget_s=15.0):
    """The oracle (a port of the reference algorithm) timed on host cores on a bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    try:
        import oracle_c
        have_c = oracle_c.available()
    except Exception:
        have_c = False
    import topsicle_oracle as orc
    pats = orc.kmer_table(motif, k)
    if have_c:
        return oracle_c.timed_baseline(seqs, pats, len(motif), prm, budget_s)
    t0 = time.perf_counter()
    done = bases = 0
    for seq in seqs:
        cs, ce = orc.trc_counts(seq, pats, prm.no_bp)
        call = orc.trc_call(cs, ce, pats, len(motif), 0.7, prm.no_bp)
        if call is not None and len(seq) > prm.min_len:
            sums = {}
            for tail in ("forward", "reverse"):        # the reference scans both tails (allsteps.py:279-291)
                _, counts = orc.window_count_matrix(seq, tail, pats, prm.window, prm.slide, prm.trimfirst, prm.maxlen)
                sums[tail] = orc.window_sums(counts)
            orc.boundary_from_sums(sums[call[1]], len(pats), prm.slide, prm.trimfirst, prm.maxlen)
        done += 1
        bases += len(seq)
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return dict(value=bases / dt, unit="bases/s", cores=1, kind="port",
                sample=f"{done} reads of the same batch, pure-Python oracle (re-free restatement of allsteps.py), "
                       f"single pass, both tails scanned like the reference; {dt:.1f} s",
                reads_per_s=done / dt)


def reference_python_baseline(bases, offsets, motif, k, prm, reads_per_core=192):
    """The reference's own CPU path beside the GPU number: oracle/ref_mirror.py -- `re.finditer` per window per
    pattern, both tails, numpy-var Binseg, the file re-parsed per passing read, one Pool task per FILE like
    Topsicle/main.py:232-235 -- on the first reads of the batch written out as one FASTQ file per usable core.  Must run
    BEFORE this process touches a GPU (the Pool forks).  The rate is extrapolated from that sample to the batch."""
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_c
    import ref_mirror
    cores = oracle_c.usable_cores()
    n = len(offsets) - 1
    per = max(1, min(reads_per_core, n // cores))
    tmp = tempfile.mkdtemp(prefix="tps_refpy_")
    paths = []
    try:
        for c in range(cores):
            pth = os.path.join(tmp, f"part{c}.fastq")
            with open(pth, "w") as h:
                for i in range(c * per, (c + 1) * per):
                    sq = bases[offsets[i]:offsets[i + 1]].tobytes().decode()
                    h.write(f"@r{i}\n{sq}\n+\n{'I' * len(sq)}\n")
            paths.append(pth)
        wall, n_pass, per_file = ref_mirror.timed_pool(paths, motif, k, prm.min_len, 0.7, prm.window, prm.slide, prm.trimfirst, prm.maxlen, cores)
    finally:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    nb = int(offsets[cores * per])
    return dict(value=nb / wall, unit="bases/s", cores=cores, kind="port", extrapolated=True, reads_per_s=cores * per / wall,
                seconds_per_read_per_core=float(np.mean(per_file)) / per,
                sample=f"{per} reads per core x {cores} cores ({n_pass} telomeric) of the same batch as {cores} FASTQ files through oracle/ref_mirror.py: "
                       f"pure-Python mirror of Topsicle's process_file (re.finditer per window per pattern, both tails, numpy-var "
                       f"Binseg, file re-parsed per passing read), one multiprocessing.Pool task per file like Topsicle/main.py:232-235; "
                       f"{wall:.1f} s wall; the rate is per-read cost x reads, i.e. extrapolated to the batch")


def gpu_topology():
    """[(pci bdf, [cpus of its NUMA node])] of the node's GPUs in KFD order, read from sysfs without touching the GPU runtime."""
    import glob
    out = []
    nodes = sorted(glob.glob("/sys/class/kfd/kfd/topology/nodes/*"), key=lambda d: int(os.path.basename(d)))
    for d in nodes:
        try:
            props = dict(l.split()[:2] for l in open(os.path.join(d, "properties")) if len(l.split()) >= 2)
        except OSError:
            continue
        if int(props.get("simd_count", "0")) == 0:                   # a CPU node
            continue
        loc, dom = int(props.get("location_id", "0")), int(props.get("domain", "0"))
        bdf = f"{dom:04x}:{(loc >> 8) & 255:02x}:{(loc >> 3) & 31:02x}.{loc & 7}"
        cpus = []
        try:
            for part in open(f"/sys/bus/pci/devices/{bdf}/local_cpulist").read().strip().split(","):
                if part:
                    lo, _, hi = part.partition("-")
                    cpus.extend(range(int(lo), int(hi or lo) + 1))
        except (OSError, ValueError):
            pass
        out.append((bdf, cpus))
    return out


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (this process has not
    touched a GPU and never will), one per GPU, rendezvous over 127.0.0.1; relay rank 0's output; non-zero exit if any
    rank fails.  Replaces the reference's Pool over input files (Topsicle/main.py:232-235) as the unit of node-level
    parallelism: one process + one context per GPU, no collective."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    topo = gpu_topology()
    share = bool(os.environ.get("TPS_BENCH_SHARE_GPU"))
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TPS_BENCH_SPAWNED="1")
        # every rank sees ONE GPU (set in the child's environment before it starts: a process that has touched a GPU is never
        # re-executed) and runs on the CPUs of that GPU's NUMA node
        if topo and (share or len(topo) >= n):
            g = r % len(topo)
            env["ROCR_VISIBLE_DEVICES"] = str(g)
            env["TPS_BENCH_DEVICE"] = "0"
            env["TPS_BENCH_PCI"] = topo[g][0]
            if topo[g][1]:
                env["TPS_BENCH_CPUS"] = ",".join(str(c) for c in topo[g][1])
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    if any(rcs):
        raise SystemExit(f"bench.py: ranks exited with {rcs}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000, help="timed steps (default 1000: ~70 ms of GPU work at config 2, long enough for an outside utilisation probe to see the GPU busy)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="config2", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end legs (FASTQ file on disk -> results / -> CLI outputs)")
    ap.add_argument("--no-store-sums", action="store_true", help="do not write S_w to HBM (boundary-only run)")
    ap.add_argument("--n-reads", type=int, default=0, help="diagnostic: override the batch size (NOT the metric's workload)")
    ap.add_argument("--k", type=int, default=0, help="diagnostic: override the k of the workload's pattern table (NOT the metric's workload)")
    ap.add_argument("--flags", type=int, default=0, help="diagnostic: override the scan flags (partial pipelines are NOT the metric)")
    ap.add_argument("--no-steady", action="store_true", help="skip the steady-state leg (the kernel on a batch four times the size: the launch ramp "
                    "and the last round of wave slots weigh a quarter; 46 launches, ~8 ms): the command whose rocprofv3 --stats average is "
                    "compared with the live kernel duration leaves it out, its launches carry the same kernel name")
    ap.add_argument("--steady", action="store_true", help=argparse.SUPPRESS)          # (round 4's opt-in switch: the leg is on by default now)
    ap.add_argument("--min-timed-ms", type=float, default=50.0, help="each timed region is repeated until this much time has been measured (and >= 3 repeats); 0 = one repeat (profiling runs)")
    ap.add_argument("--sequential-tables", action="store_true", help="several tables per step (config5): back to back on the one context instead of "
                    "overlapping on one context per table (the A/B switch; batch.SEQUENTIAL_TABLES in the pipeline)")
    ap.add_argument("--prime", type=int, default=256, help="launches before the warm-up (runtime growth steps, clocks)")
    ap.add_argument("--no-wgs", action="store_true", help="skip the e2e leg of the step-1-dominated regime")
    ap.add_argument("--streams", type=int, default=2, help="contexts (streams) the batches of the `value` region alternate between: 2 = the pipeline's "
                    "own shape (batch.EnginePool: two contexts per GPU) -- the ramp of one launch fills the drain of the previous one; 1 = strictly one "
                    "launch after the other.  The kernel's own duration and the roofline are measured on serialised launches either way")
    ap.add_argument("--resident-copies", type=int, default=0, help="copies of the batch kept in HBM (0 = enough for >1 GB)")
    ap.add_argument("--errors", default="", choices=["", "ont", "hifi", "none"], help="diagnostic: override the workload's per-base error profile (NOT the metric's workload)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus, sys.argv[1:])

    if os.environ.get("TPS_BENCH_CPUS"):         # spawned rank: the CPUs next to its GPU
        try:
            os.sched_setaffinity(0, {int(c) for c in os.environ["TPS_BENCH_CPUS"].split(",")})
        except (OSError, ValueError):
            pass
    lib = hiplib.load_library()                  # dlopen only (no GPU call): before torch.distributed brings its own HIP runtime in
    from topsicle_amd import dist
    grp = dist.Group()
    rank, world, local_rank = grp.rank, grp.world, grp.local_rank
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    cfg = dict(CONFIGS[args.workload])
    if args.n_reads:
        cfg["n_reads"] = args.n_reads
        cfg["desc"] += f" [diagnostic batch of {args.n_reads} reads]"
    if args.errors:
        cfg["errors"] = {"ont": synth.ONT, "hifi": synth.HIFI, "none": None}[args.errors]
        cfg["desc"] += f" [diagnostic error profile: {args.errors}]"
    if args.k:
        cfg["k"] = args.k
        cfg.pop("ks", None)
        cfg["desc"] += f" [diagnostic: k = {args.k}]"
    motif, k = cfg["motif"], cfg["k"]
    ks = cfg.get("ks", [k])
    tables = [kmer_table(motif, kk) for kk in ks]
    pats = tables[0]
    P = len(pats)
    # every rank scans its own, differently seeded batch of the same shape (weak scaling)
    if cfg.get("ragged"):
        bases, offsets, truth = synth.make_ragged_reads(cfg["n_reads"], motif, seed=cfg["seed"] + 1000 * rank, errors=cfg["errors"])
    else:
        bases, offsets, truth = synth.make_reads(cfg["n_reads"], cfg["read_len"], motif, seed=cfg["seed"] + 1000 * rank,
                                                 errors=cfg["errors"], telomeric_fraction=cfg.get("telomeric_fraction", 1.0))
    n_reads = cfg["n_reads"]
    batch_bases = int(offsets[-1])
    prm = hiplib.make_params(no_bp=1000, min_len=9000, min_count=min_count_for_cutoff(0.7, 1000, len(motif)),
                             window=cfg["window"], slide=cfg["slide"], trimfirst=100, maxlen=20000, jump=5, min_size=2,
                             flags=hiplib.F_STEP1 | hiplib.F_WINDOWS | hiplib.F_BINSEG |
                             (0 if args.no_store_sums else hiplib.F_STORE_SUMS))

    if cfg.get("raw"):
        prm.flags |= hiplib.F_STORE_RAW
    if args.flags:
        prm.flags = args.flags
    ref_py = None
    if not args.no_cpu_baseline and world == 1 and not (args.flags or args.n_reads or args.errors or args.k):
        try:                                      # forks a Pool: before anything touches the GPU
            ref_py = reference_python_baseline(bases, offsets, motif, k, prm)
        except Exception as e:
            ref_py = {"error": repr(e)}
    # one GPU per rank; TPS_BENCH_SHARE_GPU=1 (testing the launcher path on a box with fewer GPUs than ranks) wraps around
    import ctypes
    n_dev = ctypes.c_int(0)
    lib.tps_device_count(ctypes.byref(n_dev))
    dev = local_rank
    if "TPS_BENCH_DEVICE" in os.environ:         # spawned by this script: ROCR_VISIBLE_DEVICES leaves this rank one GPU
        dev = int(os.environ["TPS_BENCH_DEVICE"])
        if n_dev.value < 1:
            raise SystemExit(f"rank {rank}: no GPU visible ({lib.tps_last_error().decode()})")
    elif os.environ.get("TPS_BENCH_SHARE_GPU"):
        dev = local_rank % max(n_dev.value, 1)
    elif n_dev.value < world:
        raise SystemExit(f"--gpus {world} but only {n_dev.value} GPU(s) visible ({lib.tps_last_error().decode()})")
    sc = hiplib.HipScanner(dev)
    # kernel durations come from HIP events stamped by the dispatch itself; every 4th launch is timed (timing a
    # launch costs ~3.5 us of host/queue work, which would otherwise sit inside every timed step); helper contexts inherit it
    if "event_stride" not in hiplib.debug_options_from_env():
        sc.debug_option("event_stride", 4)
    # the GPU this rank really got: its PCI address from the runtime (independent of ROCR_VISIBLE_DEVICES), and -- when a launcher
    # other than this script started the rank (torch.distributed.run: no TPS_BENCH_CPUS) -- the CPUs of that GPU's NUMA node
    import re as _re
    m_pci = _re.search(r"pci=([0-9a-f]{4}:[0-9a-f]{2}:[0-9a-f]{2})", sc.device_info())
    rank_pci = os.environ.get("TPS_BENCH_PCI") or (m_pci.group(1) + ".0" if m_pci else None)
    if world > 1 and not os.environ.get("TPS_BENCH_CPUS") and rank_pci:
        try:
            cpus = set()
            for part in open(f"/sys/bus/pci/devices/{rank_pci}/local_cpulist").read().strip().split(","):
                if part:
                    lo, _, hi = part.partition("-")
                    cpus.update(range(int(lo), int(hi or lo) + 1))
            cpus &= os.sched_getaffinity(0)
            if cpus:
                os.sched_setaffinity(0, cpus)
        except (OSError, ValueError):
            pass
    sc.set_patterns(pats)
    copies = args.resident_copies or max(2, min(hiplib.MAX_SLOTS, -(-(1 << 30) // max(batch_bases, 1))))
    for s in range(copies):
        sc.upload(s, bases, offsets)
    # several tables per batch (config 5: k = 4, 5, 6): one context per table, all scanning the SAME resident batch at the same
    # time (the helpers borrow it: tps_batch_share), as batch.scan_jobs runs `--telophrase 4 5 6`;
    # --sequential-tables: back to back on the one context (the A/B switch)
    concurrent = len(tables) > 1 and not args.sequential_tables
    engines = [sc]
    if concurrent:
        engines += [sc.helper(j) for j in range(len(tables) - 1)]
        for eng, t in zip(engines, tables):
            eng.set_patterns(t)
            if eng is not sc:
                for s in range(copies):
                    eng.share(s, sc, s)

    # the longest table pass is launched first (the largest k: its launch ends last otherwise and the step with it)
    launch_order = engines[::-1]

    def sync_all():
        for eng in engines:
            eng.sync()

    def barrier():
        sync_all()
        grp.barrier()

    # prime every resident copy once (first scan of a slot plans its LDS geometry and allocates result
    # buffers), then the W untimed warm-up steps
    # ... and enough further launches (~25 ms of GPU work) that the HIP runtime's one-off internal growth steps
    # (a ~6 ms hiccup observed once around the 20th-30th launch of a process) are over and the GPU has reached its
    # sustained clocks before the timed region: after only 64 launches a step measures 0.099 ms, after 256: 0.095 ms
    def step(slot):
        if len(tables) == 1:
            sc.scan(slot, prm)
        elif concurrent:
            for eng in launch_order:               # one launch per table, each on its own stream: they overlap on the GPU
                eng.scan(slot, prm)
        else:
            for t in tables:                       # resident tables: switching is a pointer swap in the library
                sc.set_patterns(t)
                sc.scan(slot, prm)

    for s in range(max(args.prime // len(tables), 4 * copies)):
        step(s % copies)
    sync_all()
    for i in range(args.warmup):
        step(i % copies)
    import math

    def timed_region(run_steps, sync):
        """R repeats of EXACTLY args.steps steps, each bracketed by barrier + device sync on both sides and timed as the max over
        ranks; R = enough for --min-timed-ms of measured time, at least 3 (decided from the first repeat's max-over-ranks time, so
        every rank takes the same R).  Returns ([seconds of repeat r (max over ranks)], [this rank's seconds])."""
        glob_t, own_t = [], []
        repeats = 1
        r = 0
        while r < repeats:
            sync()
            grp.barrier()
            t0 = time.perf_counter()
            run_steps()
            sync()                 # device idle: every step's kernel and result copy has finished
            own = time.perf_counter() - t0
            glob_t.append(grp.max(own))
            own_t.append(own)
            grp.barrier()
            if r == 0:
                repeats = 1 if args.min_timed_ms <= 0 else int(min(400, max(3, math.ceil(args.min_timed_ms * 1e-3 / max(glob_t[0], 1e-7)))))
            r += 1
        return glob_t, own_t

    def spread(ts):
        """median / p10 / p90 of the repeats, in ms per step"""
        v = np.sort(np.asarray(ts)) / args.steps * 1e3
        return float(np.median(v)), float(v[int(0.1 * (len(v) - 1))]), float(v[int(math.ceil(0.9 * (len(v) - 1)))])

    def run_serial():
        for i in range(args.steps):
            step(i % copies)

    barrier()
    for eng in engines:
        eng.kernel_time_reset()
    serial_t, serial_own = timed_region(run_serial, sync_all)

    def median_repeat(ts):
        """index of the repeat whose max-over-ranks time is the median (the lower middle one of an even count): the headline is a
        repeat that really ran, and every rank reports its own time of THAT repeat -- `ms_per_step` stays the max over `ranks`"""
        return int(np.argsort(np.asarray(ts), kind="stable")[(len(ts) - 1) // 2])
    mid = median_repeat(serial_t)
    dt = float(serial_t[mid])
    dt_rank = float(serial_own[mid])
    if concurrent:                 # per step: the sum of the overlapping launches' own durations (each stretched by its neighbours)
        kt = [eng.kernel_time_ms() for eng in engines]
        n_launch, k_mean_ms = sum(x[0] for x in kt), sum(x[2] for x in kt) / len(tables)
    else:
        n_launch, k_total_ms, k_mean_ms = sc.kernel_time_ms()
    # The same K steps again with consecutive batches on alternating contexts (each its own stream), as the file pipeline issues them
    # (batch.EnginePool keeps two contexts per GPU): a launch's ramp fills the drain of the one before it.  This region gives `value`;
    # the serialised region above gives the kernel's own duration (the roofline) and `single_stream`.
    dt_serial = dt
    piped = None
    pipe, pipe_err = None, None
    if len(tables) == 1 and args.streams > 1:
        try:                                       # (a rank that cannot set its second context up must not leave the others in a barrier)
            pipe = [sc] + [sc.helper(j) for j in range(args.streams - 1)]
            for eng in pipe[1:]:
                eng.set_patterns(pats)
                for s in range(copies):
                    eng.share(s, sc, s)
            for i in range(max(args.warmup, 4 * len(pipe))):
                pipe[i % len(pipe)].scan(i % copies, prm)
            for eng in pipe:
                eng.sync()
        except Exception as e:
            pipe_err = repr(e)
        if grp.max(1.0 if pipe_err else 0.0) > 0.0:       # every rank or none
            pipe = None
    if pipe is not None:
        def sync_pipe():
            for eng in pipe:
                eng.sync()

        def run_piped():
            for i in range(args.steps):
                pipe[i % len(pipe)].scan(i % copies, prm)
        grp.barrier()
        for eng in pipe:
            eng.kernel_time_reset()
        piped_t, piped_own = timed_region(run_piped, sync_pipe)
        mid_p = median_repeat(piped_t)
        dt = float(piped_t[mid_p])
        ktp = [eng.kernel_time_ms() for eng in pipe]
        _med, p10, p90 = spread(piped_t)
        med = dt / args.steps * 1e3
        piped = dict(streams=len(pipe), ms_per_step=med, ms_per_step_p10=p10, ms_per_step_p90=p90, repeats=len(piped_t),
                     timed_ms_total=float(np.sum(piped_t)) * 1e3,
                     kernel_ms_mean_while_overlapping=sum(x[1] for x in ktp) / max(1, sum(x[0] for x in ktp)))
        dt_rank = float(piped_own[mid_p])
    per_rank = grp.gather_objects(dict(rank=rank, device=dev, device_info=sc.device_info(), pci=rank_pci,
                                       cpus=len(os.sched_getaffinity(0)), ms_per_step=dt_rank / args.steps * 1e3,
                                       ms_per_step_single_stream=dt_serial / args.steps * 1e3,
                                       kernel_ms_mean=k_mean_ms * len(tables), kernel_launches_timed=n_launch))
    # the same kernel on a batch four times the size (the reads repeated): the launch ramp and the partly filled last round of wave
    # slots weigh a quarter as much -- what the kernel does in steady state (rank 0 at N = 1, default workload shapes only)
    steady = None
    if not args.no_steady and world == 1 and len(tables) == 1 and not (args.flags or args.n_reads or args.errors or args.k) and copies < hiplib.MAX_SLOTS and \
            batch_bases * 4 <= (3 << 30):
        try:
            big_off = np.concatenate([offsets[:-1] + j * batch_bases for j in range(4)] + [np.array([4 * batch_bases], np.int64)])
            sc.upload(copies, np.tile(bases, 4), big_off)
            for _ in range(6):
                sc.scan(copies, prm)
            sc.sync()
            sc.kernel_time_reset()
            n_big = 40
            for _ in range(n_big):
                sc.scan(copies, prm)
            sc.sync()
            nl_big, _tot, km_big = sc.kernel_time_ms()
            steady = dict(reads_per_launch=4 * n_reads, kernel_ms_mean=km_big, kernel_launches_timed=nl_big, source="measured in this run")
        except Exception as e:                      # (out of device memory on a small GPU must not cost the bench line)
            steady = {"error": repr(e)}
    lens = np.diff(offsets)
    last = (args.steps - 1) % copies
    alg_total = alg1 = alg2 = alg3 = 0
    kinfos = []
    for j, t in enumerate(tables):                 # algorithmic bytes of every table pass of a step (one pass: the usual case)
        eng = engines[j] if concurrent else sc
        if len(tables) > 1 and not concurrent:
            sc.set_patterns(t)
            sc.scan(last, prm)
            sc.sync()
        res = eng.results(last)
        passed = res["pass"].astype(bool)
        n_win = res["n_win"].astype(np.int64)
        a_t, a_1, a_2, a_3 = algorithmic_bytes(lens, passed, n_win, len(t), prm)
        alg_total += a_t; alg1 += a_1; alg2 += a_2; alg3 += a_3
        kinfos.append(eng.kernel_info(last))
    kinfo = kinfos[0] if len(kinfos) == 1 else " | ".join(kinfos)
    k_mean_ms *= len(tables)                       # per step: the table passes' kernels together
    scanned = int((2 * np.minimum(lens, prm.no_bp)).sum() + np.maximum(np.minimum(lens, prm.maxlen) - prm.trimfirst, 0)[passed].sum())
    # bases of the batch the path can touch at all (allsteps.py:176-177, 263-271): both 1000-base heads of every read, and the
    # first / last min(L, maxlengthtelo) bases of a read that passes -- all of a read up to maxlengthtelo, 21 of 30 kb beyond
    heads = np.minimum(lens, 2 * prm.no_bp)
    tail_span = np.minimum(lens, prm.maxlen)
    touched_per_read = np.where(passed, np.minimum(lens, tail_span + np.minimum(prm.no_bp, lens - tail_span)), heads)
    touched = int(touched_per_read.sum())

    if rank == 0:
        total_bases = batch_bases * world * args.steps
        # `value`: bases the path can touch per second (= input bases while reads are no longer than maxlengthtelo: configs 1-3;
        # config 4's 30 kb reads have 9 kb nobody looks at); the plain input rate is reported beside it
        value = touched * world * args.steps / dt
        # overlapping launches: the step's wall time bounds them together (their own event durations overlap and would count
        # the GPU's time more than once); a single launch: its own duration
        roof_ms = dt / args.steps * 1e3 if concurrent else k_mean_ms
        achieved = alg_total / (roof_ms * 1e-3) / 1e9 if roof_ms > 0 else 0.0
        traffic, traffic_src = profiled_traffic(args.workload, 0, ks) if not (args.flags or args.n_reads or args.errors or args.k) else (None, None)
        _sm, s_p10, s_p90 = spread(serial_t)
        s_med = dt_serial / args.steps * 1e3
        out = {
            "metric": "bases_scanned_per_sec",
            "value": value,
            "unit": "bases/s",
            "input_bases_per_sec": total_bases / dt,
            "reads_per_sec": n_reads * world * args.steps / dt,
            "scanned_bases_per_sec": scanned * world * args.steps / dt,
            "n_gpus": world,
            "ranks": per_rank,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "ms_per_step_p10": piped["ms_per_step_p10"] if piped else s_p10,
            "ms_per_step_p90": piped["ms_per_step_p90"] if piped else s_p90,
            "repeats": piped["repeats"] if piped else len(serial_t),
            "timed_ms_total": piped["timed_ms_total"] if piped else float(np.sum(serial_t)) * 1e3,
            "streams": piped["streams"] if piped else (len(tables) if concurrent else 1),
            # strictly one launch after the other (the region the roofline's kernel duration comes from)
            "value_single_stream": touched * world * args.steps / dt_serial,
            "single_stream": {"value": touched * world * args.steps / dt_serial, "ms_per_step": s_med, "ms_per_step_p10": s_p10,
                              "ms_per_step_p90": s_p90, "repeats": len(serial_t), "timed_ms_total": float(np.sum(serial_t)) * 1e3},
            "consistency": ("`value` / `ms_per_step`: median of %d repeats of exactly %d steps with consecutive batches alternating over %d streams -- "
                            "two launches overlap, so ms_per_step may lie below the kernel's own duration (roofline.kernel_ms_mean, measured on the "
                            "serialised region = single_stream, where ms_per_step >= kernel_ms_mean holds)" % (piped["repeats"], args.steps, piped["streams"]))
                           if piped else ("`value` / `ms_per_step`: median of %d repeats of exactly %d steps, %s" %
                                          (len(serial_t), args.steps, "the step's %d table launches overlapping on %d streams" % (len(tables), len(tables))
                                           if concurrent else "strictly one launch after the other")),
            **({"pipelined_error": pipe_err} if pipe_err else {}),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": cfg["desc"],
                "reads_per_step_per_gpu": n_reads,
                "read_len": cfg["read_len"],
                "pattern": motif, "k": k if len(ks) == 1 else ks, "n_patterns": P,
                "window": cfg["window"], "slide": cfg["slide"], "trimfirst": 100, "maxlengthtelo": 20000, "cutoff": 0.7,
                "resident_copies": copies,
                "store_window_sums": not args.no_store_sums,
                "telomeric_reads_passing": int(passed.sum()),
                "scanned_bases_per_step_per_gpu": scanned,
                "sharding": "reads sharded across GPUs, one process per GPU, no collective" if world > 1 else "single GPU",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": kinfo.split()[0],
                "kernel_launch": kinfo,
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                # physical HBM rate of the same launch (PMC bytes / event time): the fused kernel moves fewer bytes than the
                # contract's algorithmic count (S_w is consumed from LDS, never re-read), so this is the lower figure
                "traffic_frac": (traffic / (roof_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (traffic and roof_ms > 0) else None,
                "algorithmic_bytes_per_launch": alg_total,
                "algorithmic_bytes_split": {"step1": alg1, "windows": alg2, "binseg": alg3},
                "kernel_ms_mean": k_mean_ms,
                "kernel_launches_timed": n_launch,
                "duration_ms": roof_ms,
                "duration_source": ("wall time of one step: %d launches (one per table) overlapping on %d streams" % (len(tables), len(tables)))
                                   if concurrent else "HIP events around the launch, serialised launches (the `single_stream` region)",
            },
        }
        roof = out["roofline"]
        if piped:
            # what the overlap recovers of the launch's ramp and drain: bytes per second over the two-stream region against the HBM peak
            # (can exceed 1: the contract's ALGORITHMIC bytes, 367 MB per config-2 launch, against the 128 MB the fused kernel really moves)
            piped["algorithmic_hbm_frac"] = alg_total / (piped["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS
            out["pipelined"] = piped
        if traffic and PROFILED_COUNTERS.get("SQ_INSTS_VALU") and roof_ms > 0:
            # what really bounds the kernel: integer VALU issue (one wave-instruction per 4 cycles per SIMD), from the committed
            # counters of the same workload; HBM carries `traffic_frac` of its peak (section 3 of DESIGN.md)
            insts = PROFILED_COUNTERS["SQ_INSTS_VALU"]                     # wave-instructions per launch (per step: all table passes)
            issue_ms = insts * 4.0 / (N_SIMDS * ENGINE_CLOCK_HZ) * 1e3      # the launch's VALU issue time with every SIMD busy
            roof["valu_busy"] = issue_ms / roof_ms
            roof["valu"] = {
                "insts_per_read": insts / n_reads,
                "lane_insts_per_base": insts * 64.0 / max(scanned * len(tables), 1),   # per base a table pass scans
                "issue_ms_per_launch": issue_ms,
                "busy_single": issue_ms / roof_ms,
                "busy_pipelined": (issue_ms / piped["ms_per_step"]) if piped else None,
                "source": "SQ_INSTS_VALU of %s x 4 cycles / (%d SIMDs x %.1f GHz) against this run's durations" % (traffic_src, N_SIMDS, ENGINE_CLOCK_HZ / 1e9),
            }
            roof["valu_busy_source"] = roof["valu"]["source"]
            roof["limiter"] = "valu-issue"
        if steady and "kernel_ms_mean" in steady and steady["kernel_ms_mean"] > 0:
            steady["frac"] = 4 * alg_total / (steady["kernel_ms_mean"] * 1e-3) / 1e9 / HBM_PEAK_GBS
            steady["workload"] = args.workload
        if steady:
            roof["steady_state"] = steady
            if "frac" in steady:
                roof["steady_state_frac"] = steady["frac"]
        if not args.no_cpu_baseline and world == 1:        # the CPU baseline leg runs at N = 1 only
            n_cpu = min(n_reads, 16384)               # (config 2: the whole batch, ~5 s on 16 cores; the budget inside stops a slow host)
            seqs = synth.split_reads(bases[: offsets[n_cpu]], offsets[: n_cpu + 1])
            out["cpu_baseline"] = cpu_baseline(seqs, motif, k, prm)
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
            if ref_py is not None:
                out["cpu_baseline"]["reference_python"] = ref_py
        if not args.no_e2e and world == 1 and not (args.flags or args.n_reads):
            # PCIe- and parse-inclusive rates on a FASTQ file of the same reads: never `value` (topsicle_amd/e2e.py)
            from topsicle_amd import e2e
            try:
                out["e2e"] = e2e.measure(bases, offsets, motif, k, cfg["slide"], device=dev)
            except Exception as e:                  # a full /tmp must not cost the bench line
                out["e2e"] = {"error": repr(e)}
            if args.workload == "config2" and not args.no_wgs:
                try:                                # the step-1-dominated regime end to end: 30 kb reads, 1 % telomeric, one-pass against two-pass upload
                    out["e2e"]["wgs_1pct"] = e2e.measure_wgs(device=dev)
                except Exception as e:
                    out["e2e"]["wgs_1pct"] = {"error": repr(e)}
        print(json.dumps(out))
    sc.close()
    grp.close()


if __name__ == "__main__":
    main()
