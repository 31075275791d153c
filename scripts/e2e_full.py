#!/usr/bin/env python3
"""BASELINE configs[2..4] at their stated PER-GPU size, file -> CLI outputs (not part of the default bench).

For every config one GPU's shard is generated (synthetic reads of the config's shape), written as plain FASTQ, ordinary gzip
(level 1, one stream) and BGZF (bgzip's layout), and the `topsicle` CLI (topsicle_amd.main) is run on each file as a child
process.  Recorded per run: wall seconds, input bases per second (ONE run each: these are minutes-long, not best-of-n), peak
and sampled resident memory (flat or growing?), pinned staging bytes, bytes written, and a row-level spot check of
telolengths_all.csv against the C oracle (oracle/oracle.c -- the checker, here as in tests/).

    python scripts/e2e_full.py [--scale 1.0] [--configs 2,3,4] [--formats fastq,gz,bgzf] [--workdir DIR] [--out profiles/r04_x/e2e_full.json]

--scale shrinks the read counts (the default 1.0 = 25 000 x 20 kb, 125 000 x 30 kb, 62 500 x 25 kb); the script shrinks it by
itself when the work directory has too little free space and says so in the output.
Replaces, at scale: Topsicle/main.py:52-154 (process_file), 206-235 (the loop over k and files)."""
import argparse
import csv
import json
import os
import resource
import shutil
import sys
import threading
import time
import zlib
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import numpy as np  # noqa: E402

from topsicle_amd import e2e, synth  # noqa: E402  (no GPU call in this process: the CLI runs as a child)

CONFIGS = {
    2: dict(name="configs[2] shard", n_reads=25000, read_len=20000, motif="AAACCCT", errors=synth.HIFI, seed=20250919 + 2,
            cli=["--pattern", "AAACCCT"], ks=[5], slide=7, cutoff=0.7),
    3: dict(name="configs[3] shard", n_reads=125000, read_len=30000, motif="CCCTAA", errors=synth.ONT, seed=20250919 + 3,
            cli=["--pattern", "CCCTAA", "--cutoff", "0.3", "0.4", "0.5", "0.6", "0.7", "0.8"], ks=[4], slide=6, cutoff=0.3),
    4: dict(name="configs[4] shard", n_reads=62500, read_len=25000, motif="CCCTAA", errors=synth.ONT, seed=20250919 + 4,
            cli=["--pattern", "CCCTAA", "--telophrase", "4", "5", "6", "--rawcountpattern", "--rawcountformat", "npz"], ks=[4, 5, 6], slide=6,
            cutoff=0.7),
}


def rss_mb():
    for ln in open("/proc/self/status"):
        if ln.startswith("VmRSS:"):
            return int(ln.split()[1]) / 1024.0
    return 0.0


class RssSampler(threading.Thread):
    def __init__(self, period=0.25):
        super().__init__(daemon=True)
        self.period, self.samples, self.stop_ev = period, [], threading.Event()

    def run(self):
        t0 = time.perf_counter()
        while not self.stop_ev.is_set():
            self.samples.append((time.perf_counter() - t0, rss_mb()))
            self.stop_ev.wait(self.period)

    def summary(self):
        if not self.samples:
            return {}
        v = np.array([s[1] for s in self.samples])
        q = [float(v[int(f * (len(v) - 1))]) for f in (0.25, 0.5, 0.75, 1.0)]
        return {"rss_mb_peak": round(float(v.max()), 1), "rss_mb_at_25_50_75_100_pct_of_the_run": [round(x, 1) for x in q], "rss_samples": len(v)}


def write_gzip_stream(dst, src, level=1):
    co = zlib.compressobj(level, zlib.DEFLATED, 31)
    with open(src, "rb") as s, open(dst, "wb") as d:
        while True:
            blk = s.read(32 << 20)
            if not blk:
                break
            d.write(co.compress(blk))
        d.write(co.flush())


def write_bgzf_parallel(dst, src, block=65280, level=1, threads=16):
    import struct

    def member(chunk):
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        body = co.compress(chunk) + co.flush()
        bsize = 12 + 6 + len(body) + 8
        return (b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1) +
                body + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
    with open(src, "rb") as s, open(dst, "wb") as d, ThreadPoolExecutor(threads) as ex:
        while True:
            big = s.read(block * 512)
            if not big:
                break
            for m in ex.map(member, [big[i:i + block] for i in range(0, len(big), block)]):
                d.write(m)
        d.write(member(b""))


def dir_bytes(path):
    return sum(os.path.getsize(os.path.join(r, f)) for r, _d, fs in os.walk(path) for f in fs)


def spot_check(cfg, bases, offsets, outdir, n_check=48):
    """Rows of telolengths_all.csv (first k) against oracle.c on the same reads: every read's pass / TRC / boundary for a sample."""
    import oracle_c
    import topsicle_oracle as orc
    k = cfg["ks"][0]
    pats = orc.kmer_table(cfg["motif"], k)
    rows = {}
    for r in csv.reader(open(os.path.join(outdir, "telolengths_all.csv"))):
        if r and r[0] != "file_name" and r[1] == str(k):
            rows[r[3]] = r
    n = len(offsets) - 1
    idx = sorted(set(np.linspace(0, n - 1, n_check).astype(int).tolist()))
    sub_off = np.zeros(len(idx) + 1, np.int64)
    np.cumsum([int(offsets[i + 1] - offsets[i]) for i in idx], out=sub_off[1:])
    sub = np.concatenate([bases[offsets[i]:offsets[i + 1]] for i in idx])
    out, done, _ = oracle_c.batch(sub, sub_off, pats, len(cfg["motif"]), 1000, 9000, cfg["cutoff"], 100, cfg["slide"], 100, 20000,
                                  both_tails=False, threads=min(16, oracle_c.usable_cores()))
    assert done == len(idx)
    bad = []
    for j, i in enumerate(idx):
        rid = f"read{i}"
        if out[j, 0]:
            trc = out[j, 3] / (1000 / len(cfg["motif"]))
            want = [f"{trc:.3f}", rid, str(int(out[j, 6]))]
            if rid not in rows or rows[rid][2:] != want:
                bad.append((rid, rows.get(rid), want))
        elif rid in rows:
            bad.append((rid, rows[rid], None))
    rep = {"reads_checked": len(idx), "rows_in_csv_for_first_k": len(rows), "mismatches": len(bad), "first_mismatches": bad[:3]}
    # raw rows (--rawcountformat npz): every archive's member CRCs (zipfile.testzip re-reads it once), and for a sample of reads the
    # rows themselves against the oracle -- mapped in place (counts.npy is stored, its rows begin at byte 4096: topsicle_amd/rawnpz.py)
    import glob
    import zipfile
    raw_bad, raw_checked, crc_ok = 0, 0, True
    for path in sorted(glob.glob(os.path.join(outdir, "rawcount_*.npz"))):
        kk = int(os.path.basename(path).split("_")[1])
        with zipfile.ZipFile(path) as zf:
            crc_ok = crc_ok and zf.testzip() is None
            ids = np.load(zf.open("read_id.npy")).tolist()
            tails = np.load(zf.open("tail.npy")).tolist()
            wo = np.load(zf.open("win_off.npy"))
        pk = orc.kmer_table(cfg["motif"], kk)
        rows_mm = np.memmap(path, dtype=np.uint8, mode="r", offset=4096, shape=(int(wo[-1]), len(pk)))
        pos = {r: j for j, r in enumerate(ids)}
        for i in idx[::6]:
            j = pos.get(f"read{i}")
            if j is None:
                continue
            seq = bases[offsets[i]:offsets[i + 1]].tobytes().decode()
            want = orc.window_count_matrix(seq, tails[j], pk, 100, cfg["slide"], 100, 20000)[1]
            raw_checked += 1
            raw_bad += not np.array_equal(rows_mm[wo[j]:wo[j + 1]], want)
        del rows_mm
    if raw_checked or not crc_ok:
        rep.update({"raw_rows_reads_checked": raw_checked, "raw_rows_mismatches": raw_bad, "npz_member_crcs_ok": crc_ok})
    return rep


def run_cli(path, outdir, cfg, device):
    """`python -m topsicle_amd.main ...` as a child process (a fresh process per run, as a user starts it: its resident memory is
    sampled from /proc while it runs)."""
    import subprocess
    shutil.rmtree(outdir, ignore_errors=True)
    argv = [sys.executable, "-m", "topsicle_amd.main", "--inputDir", path, "--outputDir", outdir, "--device", str(device)] + cfg["cli"]
    samples = []
    t0 = time.perf_counter()
    os.makedirs(os.path.dirname(outdir), exist_ok=True)
    with open(outdir + ".stdout", "w") as so, open(outdir + ".stderr", "w") as se:
        p = subprocess.Popen(argv, cwd=ROOT, stdout=so, stderr=se)
        while p.poll() is None:
            try:
                for ln in open(f"/proc/{p.pid}/status"):
                    if ln.startswith("VmRSS:"):
                        samples.append((time.perf_counter() - t0, int(ln.split()[1]) / 1024.0))
                        break
            except OSError:
                pass
            time.sleep(0.2)
    wall = time.perf_counter() - t0
    if p.returncode != 0:
        raise RuntimeError(f"topsicle exited with {p.returncode}: {open(outdir + '.stderr').read()[-2000:]}")
    d = {"wall_s": round(wall, 3), "output_bytes": dir_bytes(outdir)}
    for ln in open(outdir + ".stdout"):
        if ln.startswith("Elapsed time(s):"):
            d["cli_elapsed_s"] = float(ln.split()[2])            # the CLI's own clock (without the interpreter's start-up)
    os.unlink(outdir + ".stdout")
    os.unlink(outdir + ".stderr")
    if samples:
        v = np.array([x[1] for x in samples])
        d.update({"rss_mb_peak": round(float(v.max()), 1), "rss_samples": len(v),
                  "rss_mb_at_25_50_75_100_pct_of_the_run": [round(float(v[int(f * (len(v) - 1))]), 1) for f in (0.25, 0.5, 0.75, 1.0)]})
        d["rss_series_s_mb"] = [[round(t, 2), round(m)] for t, m in samples]       # (one sample per 0.2 s: where a peak sits in the run)
    log = os.path.join(outdir, "topsicle_run.log")
    if os.path.exists(log):
        d["two_pass_line"] = [ln.strip() for ln in open(log) if "two passes" in ln][:1]
        d["writer_line"] = [ln.strip().split("] ", 1)[-1] for ln in open(log) if "writer thread" in ln][:3]     # where the filtered file's time went
        d["shards_line"] = [ln.strip().split("] ", 1)[-1] for ln in open(log) if "byte ranges" in ln][:1]
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--configs", default="2,3,4")
    ap.add_argument("--formats", default="fastq,gz,bgzf")
    ap.add_argument("--workdir", default=None)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    work = args.workdir or ("/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > (64 << 30) else os.environ.get("TMPDIR") or "/tmp")
    os.makedirs(work, exist_ok=True)
    report = {"host_cpus_usable": len(os.sched_getaffinity(0)), "workdir": work, "runs": []}
    for c in [int(x) for x in args.configs.split(",")]:
        cfg = dict(CONFIGS[c])
        n = max(64, int(cfg["n_reads"] * args.scale))
        need = 2.0 * n * cfg["read_len"] * 2.6                     # text + compressed copies + the filtered output of one run
        free = shutil.disk_usage(work).free
        shrunk = 1.0
        if need > 0.8 * free:
            shrunk = 0.8 * free / need
            n = max(64, int(n * shrunk))
        tmp = os.path.join(work, f"tps_e2e_full_c{c}")
        shutil.rmtree(tmp, ignore_errors=True)
        os.makedirs(tmp)
        try:
            t0 = time.perf_counter()
            bases, offsets, _ = synth.make_reads(n, cfg["read_len"], cfg["motif"], seed=cfg["seed"], errors=cfg["errors"])
            t_gen = time.perf_counter() - t0
            n_bases = int(offsets[-1])
            fq = os.path.join(tmp, "shard.fastq")
            t0 = time.perf_counter()
            e2e.write_fastq(fq, bases, offsets)
            t_wr = time.perf_counter() - t0
            files = {"fastq": fq}
            entry = {"config": cfg["name"], "reads": n, "read_len": cfg["read_len"], "bases": n_bases, "cli_args": cfg["cli"],
                     "scale": args.scale, "shrunk_for_disk_space_by": round(shrunk, 3), "generate_s": round(t_gen, 1), "write_fastq_s": round(t_wr, 1),
                     "fastq_bytes": os.path.getsize(fq), "formats": {}}
            for fmt in args.formats.split(","):
                if fmt == "gz":
                    files[fmt] = os.path.join(tmp, "shard_gz.fastq.gz")
                    t0 = time.perf_counter()
                    write_gzip_stream(files[fmt], fq)
                    entry["formats"].setdefault(fmt, {})["compress_s"] = round(time.perf_counter() - t0, 1)
                elif fmt == "bgzf":
                    files[fmt] = os.path.join(tmp, "shard_bgzf.fastq.gz")
                    t0 = time.perf_counter()
                    write_bgzf_parallel(files[fmt], fq)
                    entry["formats"].setdefault(fmt, {})["compress_s"] = round(time.perf_counter() - t0, 1)
                elif fmt != "fastq":
                    continue
                outdir = os.path.join(tmp, "out_" + fmt)
                run = run_cli(files[fmt], outdir, cfg, args.device)
                run["input_file_bytes"] = os.path.getsize(files[fmt])
                run["bases_per_s"] = n_bases / run.get("cli_elapsed_s", run["wall_s"])
                run["spot_check_vs_oracle_c"] = spot_check(cfg, bases, offsets, outdir)
                run["outputs"] = sorted(os.listdir(outdir))[:12]
                entry["formats"].setdefault(fmt, {}).update(run)
                print(json.dumps({"config": c, "format": fmt, **{k: run[k] for k in ("wall_s", "bases_per_s", "rss_mb_peak", "output_bytes")},
                                  "mismatches": run["spot_check_vs_oracle_c"]["mismatches"]}), flush=True)
                shutil.rmtree(outdir, ignore_errors=True)
                if fmt != "fastq":
                    os.unlink(files[fmt])
            report["runs"].append(entry)
            del bases
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    report["max_rss_mb_of_the_process"] = round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0, 1)
    txt = json.dumps(report, indent=1, default=str)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        open(args.out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
