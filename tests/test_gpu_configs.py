"""GPU parity at the shapes of BASELINE.json's configs that the first suite did not reach:

  configs[0]  demo file, --pattern AAACCCT, defaults (slide = len(pattern) = 7)
  configs[2]  PacBio-HiFi-like reads x 20 kb, AAACCCT k=5 slide 7 (2829 windows, kernel _s7)
  configs[3]  ONT-like reads x 30 kb (L > maxlengthtelo), CCCTAA k=4 slide 6 (3301 windows, HALO tiles), --cutoff sweep
  every config: change-point == the FLOAT64 restatement of what allsteps.py:310-311 runs (oracle.c, numpy order),
                over thousands of reads, not only the exact-rational arg-max

Everything goes through the C ABI on a real MI355X.  Integer work is compared bit-exact.
"""
import csv
import json
import os

import numpy as np
import pytest

import oracle_c
import topsicle_oracle as orc
from topsicle_amd import allsteps, hiplib, synth

pytestmark = pytest.mark.gpu
TAILS = ["forward", "reverse"]
FULL = hiplib.F_STEP1 | hiplib.F_WINDOWS | hiplib.F_BINSEG | hiplib.F_STORE_SUMS


@pytest.fixture(scope="module")
def sc():
    s = hiplib.HipScanner(0)
    yield s
    s.close()


def _params(motif, slide, cutoff=0.7, min_len=9000, flags=FULL):
    return hiplib.make_params(no_bp=1000, min_len=min_len, min_count=allsteps.min_count_for_cutoff(cutoff, 1000 / len(motif), 1000),
                              window=100, slide=slide, trimfirst=100, maxlen=20000, flags=flags)


def _scan(sc, slot, bases, offsets, prm):
    sc.upload(slot, bases, offsets)
    sc.scan(slot, prm)
    sc.sync()
    res = sc.results(slot).copy()
    sums, win_off = sc.window_sums(slot)
    return res, sums, win_off


def _check_reads_vs_oracle(bases, offsets, res, sums, win_off, pats, motif, slide, cutoff, idx):
    for i in idx:
        seq = bytes(bases[offsets[i]:offsets[i + 1]]).decode()
        cs, ce = orc.trc_counts(seq, pats)
        call = orc.trc_call(cs, ce, pats, len(motif), cutoff)
        assert bool(res["pass"][i]) == (call is not None and len(seq) > 9000), i
        if not res["pass"][i]:
            continue
        assert TAILS[int(res["tail"][i])] == call[1]
        s_c, _ = oracle_c.window_counts(seq, call[1], pats, 100, slide, 100, 20000)
        assert np.array_equal(sums[win_off[i]:win_off[i + 1]], s_c), i
        want, _ = oracle_c.binseg_l2(s_c, len(pats))                 # float64, numpy order: what ruptures computes
        assert int(res["bkp"][i]) == (-1 if want is None else want), i


def _float64_pipeline(bases, offsets, pats, motif, slide, cutoff):
    """Every read through oracle.c: rows [pass, tail, best_idx, best_count, n_win, bkp (float64 Binseg), boundary_bp] and, per read,
    the checksum of its window sums (round 4: the at-scale comparisons cover EVERY window of every read, not a sample)."""
    out, ck = oracle_c.batch_ck(bases, offsets, pats, len(motif), 1000, 9000, cutoff, 100, slide, 100, 20000,
                                both_tails=False, threads=oracle_c.usable_cores())
    return out, ck


def _assert_equals_float64_pipeline(res, out_ck, sums=None, win_off=None):
    out, ck = out_ck
    if sums is not None:
        p_ = res["pass"].astype(bool)
        got = oracle_c.checksums(sums, win_off)
        bad = np.nonzero(got[p_] != ck[p_, 0])[0]
        print(f"window sums: GPU vs oracle.c checksums on {int(p_.sum())} reads ({int(np.diff(win_off)[p_].sum())} windows): {len(bad)} differ")
        assert len(bad) == 0, bad[:10]
    assert np.array_equal(res["pass"], out[:, 0])
    p = res["pass"].astype(bool)
    assert np.array_equal(res["tail"][p], out[p, 1])
    best = np.where(res["tail"] == 0, res["best_start"], res["best_end"])
    assert np.array_equal(best[p], out[p, 3])
    assert np.array_equal(res["n_win"][p], out[p, 4])
    differ = np.nonzero(res["bkp"][p] != out[p, 5])[0]
    print(f"change-point: GPU vs float64 restatement on {int(p.sum())} reads: {len(differ)} disagreements")
    assert len(differ) == 0, (differ[:10], res["bkp"][p][differ[:10]], out[p, 5][differ[:10]])


# --------------------------------------------------------------------------------------------- configs[2]
def test_config3_hifi_20kb_slide7_vs_oracle(sc):
    motif, k, slide = "AAACCCT", 5, 7
    pats = orc.kmer_table(motif, k)
    sc.set_patterns(pats)
    bases, offsets, truth = synth.make_reads(64, 20000, motif, seed=20250919 + 2, errors=synth.HIFI)
    res, sums, win_off = _scan(sc, 0, bases, offsets, _params(motif, slide))
    assert "tps_scan_kernel_s7" in sc.kernel_info(0)
    assert res["pass"].all() and (res["n_win"] == 2829).all()
    assert np.array_equal(res["tail"], truth["reverse"].astype(np.int32))
    _check_reads_vs_oracle(bases, offsets, res, sums, win_off, pats, motif, slide, 0.7, range(0, 64, 4))
    # HiFi-like error rates: the boundary sits on the planted tract end (the split lands about half a window before it)
    called = res["bkp"].astype(np.int64) * slide + 100
    assert np.median(np.abs(called - truth["tract"])) <= 80


def test_config3_full_shard_properties_and_float64_binseg(sc):
    """25 000 HiFi reads x 20 kb = one GPU's shard of BASELINE configs[2]: strand symmetry window by window, and the
    change-point of every read against the float64 restatement of the reference's Binseg."""
    motif, k, slide = "AAACCCT", 5, 7
    pats = orc.kmer_table(motif, k)
    sc.set_patterns(pats)
    n, L, nw = 25000, 20000, 2829
    bases, offsets, truth = synth.make_reads(n, L, motif, seed=20250919 + 2, errors=synth.HIFI)
    prm = _params(motif, slide)
    res, sums, win_off = _scan(sc, 0, bases, offsets, prm)
    assert res["pass"].mean() > 0.999 and (res["n_win"][res["pass"] == 1] == nw).all()
    comp = np.zeros(256, np.uint8)
    comp[list(b"ACGT")] = list(b"TGCA")
    rc = comp[bases.reshape(n, L)[:, ::-1]].reshape(-1)
    res_rc, sums_rc, _ = _scan(sc, 1, rc, offsets, prm)
    strict = (res["best_start"] != res["best_end"]) & (res["pass"] == 1)
    assert strict.mean() > 0.99
    assert np.array_equal(res["best_start"], res_rc["best_end"]) and np.array_equal(res["best_end"], res_rc["best_start"])
    assert np.array_equal(res["tail"][strict], 1 - res_rc["tail"][strict])
    S, S_rc = sums.reshape(n, nw), sums_rc.reshape(n, nw)
    assert np.array_equal(S[strict], S_rc[strict])
    assert np.array_equal(res["bkp"][strict], res_rc["bkp"][strict])
    assert sums.min() >= len(pats) and sums.max() <= len(pats) * (99 // k)
    # float64 Binseg (oracle.c) on a 4000-read slice of the same batch
    m = 4000
    out = _float64_pipeline(bases[: offsets[m]], offsets[: m + 1], pats, motif, slide, 0.7)
    _assert_equals_float64_pipeline(res[:m], out, sums, win_off[: m + 1])


# --------------------------------------------------------------------------------------------- configs[3]
def test_config4_ont_30kb_vs_oracle_and_float64_binseg(sc):
    """30 kb reads, 20 kb scanned (L > maxlengthtelo): 3301 windows in 7 tiles of the sums-only pair kernel."""
    motif, k, slide = "CCCTAA", 4, 6
    pats = orc.kmer_table(motif, k)
    sc.set_patterns(pats)
    n = 3000
    bases, offsets, truth = synth.make_reads(n, 30000, motif, seed=20250919 + 3)
    cutoff = 0.3                                              # min of the config's sweep 0.3 .. 0.8
    res, sums, win_off = _scan(sc, 0, bases, offsets, _params(motif, slide, cutoff))
    info = sc.kernel_info(0)
    assert "tps_scan_kernel_s6p " in info + " ", info         # the sums-only pair-table kernel
    assert (res["n_win"][res["pass"] == 1] == 3301).all()
    _check_reads_vs_oracle(bases, offsets, res, sums, win_off, pats, motif, slide, cutoff, range(0, n, 47))
    out = _float64_pipeline(bases, offsets, pats, motif, slide, cutoff)
    _assert_equals_float64_pipeline(res, out, sums, win_off)
    # the sweep itself is a host-side threshold on the step-1 counts: same reads as a scan at each cutoff
    best = np.where(res["tail"] == 0, res["best_start"], res["best_end"])
    for c in (0.3, 0.4, 0.5, 0.6, 0.7, 0.8):
        sc.scan(0, _params(motif, slide, c, flags=hiplib.F_STEP1))
        sc.sync()
        r = sc.results(0)
        assert np.array_equal(r["pass"] == 1, best / (1000 / 6) > c)


def test_config2_float64_binseg_at_scale(sc):
    """BASELINE configs[1] at full size (10 000 x 15 kb): every read's boundary against the float64 restatement."""
    motif, k, slide = "CCCTAA", 4, 6
    pats = orc.kmer_table(motif, k)
    sc.set_patterns(pats)
    bases, offsets, _ = synth.make_reads(10000, 15000, motif, seed=20250919 + 1)
    res, sums, win_off = _scan(sc, 0, bases, offsets, _params(motif, slide))
    assert "tps_scan_kernel_s6p " in sc.kernel_info(0) + " "
    out = _float64_pipeline(bases, offsets, pats, motif, slide, 0.7)
    _assert_equals_float64_pipeline(res, out, sums, win_off)


@pytest.mark.parametrize("k", [5, 6])
def test_config5_float64_binseg_self_overlap_tables(sc, k):
    motif, slide = "CCCTAA", 6
    pats = orc.kmer_table(motif, k)
    sc.set_patterns(pats)
    bases, offsets, _ = synth.make_reads(2000, 25000, motif, seed=20250919 + 4)
    res, sums, win_off = _scan(sc, 0, bases, offsets, _params(motif, slide))
    out = _float64_pipeline(bases, offsets, pats, motif, slide, 0.7)
    _assert_equals_float64_pipeline(res, out, sums, win_off)


def _write_fastq(path, bases, offsets, prefix="r"):
    with open(path, "wb") as h:
        for i in range(len(offsets) - 1):
            s = bytes(bases[offsets[i]:offsets[i + 1]])
            h.write(b"@%s%d sample\n" % (prefix.encode(), i) + s + b"\n+\n" + b"I" * len(s) + b"\n")


def test_config4_cli_cutoff_sweep_on_gpu(tmp_path):
    """`topsicle --cutoff 0.3 .. 0.8` on 30 kb reads with tracts around the 1 kb head: reads are FILTERED at min(cutoff)
    and the summary REPORTS cutoff[0] (main.py:56, 254-257)."""
    from topsicle_amd import main as cli
    motif = "CCCTAA"
    pats = orc.kmer_table(motif, 4)
    bases, offsets, _ = synth.make_reads(160, 30000, motif, seed=77, tract_min=150, tract_max=1400)
    fq = tmp_path / "ont.fastq"
    _write_fastq(fq, bases, offsets)
    out = tmp_path / "out"
    cli.main(["-i", str(fq), "-o", str(out), "--pattern", motif, "--cutoff", "0.8", "0.3", "0.4", "0.5", "0.6", "0.7"])
    rows = list(csv.reader(open(out / "telolengths_all.csv")))[1:]
    want = []
    for i in range(160):
        seq = bytes(bases[offsets[i]:offsets[i + 1]]).decode()
        cs, ce = orc.trc_counts(seq, pats)
        call = orc.trc_call(cs, ce, pats, 6, 0.3)
        if call is not None:
            want.append([f"{call[2]:.3f}", f"r{i}", str(orc.step2(seq, call[1], pats, 100, 6, 100, 20000))])
    assert 40 < len(want) < 160                                # the sweep's lowest cutoff really filters
    assert [r[2:] for r in rows] == want
    assert any(float(r[2]) < 0.8 for r in rows)                # kept although below cutoff[0]
    assert os.path.exists(out / "ont_trc_over_0.3.fastq")
    assert "with TRC >= 0.8" in open(out / "topsicle_run.log").read()


@pytest.mark.gpu
def test_cli_several_files_concurrently_equals_one_at_a_time(tmp_path):
    """Six plain FASTQ files (the packed reader's thread team, the detached unmapping and two contexts per GPU are shared by
    concurrent file threads), `--threads 4` against `--threads 1`: same rows per file, same filtered files."""
    from topsicle_amd import main as cli
    motif = "CCCTAA"
    indir = tmp_path / "in"
    indir.mkdir()
    for f in range(6):
        bases, offsets, _ = synth.make_reads(700 + 150 * f, 9000 + 1500 * f, motif, seed=100 + f, tract_min=200, tract_max=4000)
        _write_fastq(indir / f"s{f}.fastq", bases, offsets)
    outs = []
    for threads in ("4", "1"):
        out = tmp_path / f"out{threads}"
        cli.main(["-i", str(indir), "-o", str(out), "--pattern", motif, "--threads", threads, "--minSeqLength", "5000"])
        rows = sorted(tuple(r) for r in list(csv.reader(open(out / "telolengths_all.csv")))[1:])
        filt = {n: open(out / n, "rb").read() for n in sorted(os.listdir(out)) if "_trc_over_" in n}
        outs.append((rows, filt))
    assert len(outs[0][0]) > 1500 and outs[0][0] == outs[1][0]
    assert list(outs[0][1]) == list(outs[1][1]) and len(outs[0][1]) == 6
    for n in outs[0][1]:
        assert outs[0][1][n] == outs[1][1][n], n


# --------------------------------------------------------------------------------------------- configs[0]
def test_config1_demo_default_slide7_on_gpu(sc, tmp_path, gold_dir, demo_records, demo_windows):
    """BASELINE configs[0] as stated: --pattern AAACCCT, defaults (slide 7).  The k-mer set equals CCCTAAA's, so the
    step-1 rows equal the shipped CSV's; the slide-7 window matrices come from the reference itself (s7_* goldens)."""
    import shutil
    from topsicle_amd import main as cli
    meta, arrs = demo_windows
    seqs = dict(demo_records)
    pats = orc.kmer_table("AAACCCT", 5)
    assert pats == meta["patterns"]
    sc.set_patterns(pats)
    for m in meta["reads"]:
        if m.get("slide") != 7:
            continue
        bases, offsets = hiplib.pack_reads([seqs[m["id"]]])
        tail = TAILS.index(m["tail"])
        sums, win_off, raw = sc.window_counts(bases, offsets, [tail], 100, 7, 100, 20000, raw=True)
        want = arrs[f"s7_counts_{m['key'][3:]}"]
        assert np.array_equal(raw, want) and np.array_equal(sums, want.sum(axis=1, dtype=np.int64))
        bkp, _ = sc.binseg_l2(sums, win_off, len(pats))
        assert int(bkp[0]) * 7 + 100 == m["boundary"]
    d = tmp_path / "in"
    d.mkdir()
    shutil.copyfile(os.path.join(gold_dir, "demo_col0.fastq.gz"), d / "Col-0-6909_GWHBDNP00000001.1_nano_right.fastq.gz")
    out = tmp_path / "out"
    cli.main(["--inputDir", str(d), "--outputDir", str(out), "--pattern", "AAACCCT", "--threads", "1"])
    rows = list(csv.reader(open(out / "telolengths_all.csv")))[1:]
    gold = list(csv.reader(open(os.path.join(gold_dir, "demo_telolengths_all.csv"))))[1:]
    assert [r[:4] for r in rows] == [g[:4] for g in gold]      # same reads, same TRC (identical k-mer set)
    step1 = {r[0]: r for r in json.load(open(os.path.join(gold_dir, "demo_step1.json")))[2]["rows"]}
    for r in rows:
        assert step1[r[3]][2] in TAILS
        want = orc.step2(seqs[r[3]], step1[r[3]][2], pats, 100, 7, 100, 20000)
        assert int(r[4]) == want
    by_id = {r[3]: int(r[4]) for r in rows}
    for m in meta["reads"]:
        if m.get("slide") == 7 and m["tail"] == step1[m["id"]][2]:
            assert by_id[m["id"]] == m["boundary"]             # the reference's own slide-7 boundary


# --------------------------------------------------------------------------------------------- multi-rank launcher
def _run_bench(extra_args, extra_env):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TPS_BENCH_PRIME="8", **extra_env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "4", "--warmup", "1", "--no-cpu-baseline",
                           "--n-reads", "2000", "--resident-copies", "2"] + extra_args, env=env, capture_output=True, text=True, timeout=600)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: two child ranks (sharing the one GPU of this box), one
    JSON line with n_gpus = 2 and both ranks' timings; without the sharing switch the missing second GPU is an error."""
    r = _run_bench(["--gpus", "2"], {"TPS_BENCH_SHARE_GPU": "1"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and sorted(x["rank"] for x in out["ranks"]) == [0, 1]
    # `value` comes from the two-stream region, the kernel's own duration from the serialised one
    assert out["streams"] == 2 and out["pipelined"]["streams"] == 2 and out["single_stream"]["ms_per_step"] > 0
    assert out["ms_per_step"] == pytest.approx(out["pipelined"]["ms_per_step"])
    assert out["value"] > 0 and out["roofline"]["kernel"].startswith("tps_scan_kernel")
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    topo = bench.gpu_topology()                                  # sysfs only: no GPU runtime call
    for x in out["ranks"]:                                       # an N-rank line must be interpretable rank by rank
        assert x["kernel_ms_mean"] > 0 and x["kernel_launches_timed"] > 0 and "gfx950" in x["device_info"] and "pci=" in x["device_info"]
        if topo:                                                 # every spawned rank is handed ONE GPU (ROCR_VISIBLE_DEVICES) and sees it as device 0
            assert x["pci"] in [t[0] for t in topo] and x["device"] == 0
    n_dev = hiplib.load_library()
    import ctypes
    cnt = ctypes.c_int(0)
    n_dev.tps_device_count(ctypes.byref(cnt))
    if cnt.value < 2:
        r = _run_bench(["--gpus", "2"], {})
        assert r.returncode != 0 and "GPU(s) visible" in (r.stdout + r.stderr)


@pytest.mark.parametrize("workload,extra", [("config3_per_gpu", ["--n-reads", "1500"]), ("config4_sample", ["--n-reads", "1200"]), ("config5", ["--n-reads", "600"])])
def test_bench_ranks_on_the_other_workloads(workload, extra):
    """The self-launched multi-rank path with the other BASELINE shapes (HiFi slide 7, 30 kb reads, the three-k raw-count step):
    two ranks sharing this box's GPU."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TPS_BENCH_PRIME="4", TPS_BENCH_SHARE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-e2e",
                        "--resident-copies", "2", "--workload", workload, "--gpus", "2"] + extra, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 2 and len(out["ranks"]) == 2 and out["value"] > 0
    assert all(x["kernel_ms_mean"] > 0 for x in out["ranks"])


def test_bench_eight_ranks_through_the_drivers_launcher():
    """The driver's own N = 8 command -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1
    ... bench.py --gpus 8` -- on this box's ONE GPU (TPS_BENCH_SHARE_GPU=1: the ranks wrap around the visible devices): the
    eight-rank gloo rendezvous, the barrier-bracketed timing, the max over ranks and the one JSON line with all eight ranks'
    own numbers.  (What eight GPUs give is the driver's to measure; that the command runs is checked here.)"""
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, TPS_BENCH_PRIME="8", TPS_BENCH_SHARE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "2",
                        "--resident-copies", "2"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                       # rank 0 prints, once
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["steps"] == 20 and out["warmup"] == 2 and out["scaling"] == "weak"
    assert sorted(x["rank"] for x in out["ranks"]) == list(range(8))
    assert all(x["kernel_ms_mean"] > 0 and x["ms_per_step"] > 0 for x in out["ranks"])
    assert abs(out["ms_per_step"] - max(x["ms_per_step"] for x in out["ranks"])) < 1e-9      # the MAX over ranks
    assert "cpu_baseline" not in out and "e2e" not in out        # N = 1 legs only
    # whole-job value: eight ranks' batches over the slowest rank's time
    per_step = out["config"]["reads_per_step_per_gpu"] * out["config"]["read_len"]
    assert out["input_bases_per_sec"] == pytest.approx(8 * per_step / (out["ms_per_step"] * 1e-3), rel=0.02)
