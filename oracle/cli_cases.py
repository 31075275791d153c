"""TEST INFRASTRUCTURE ONLY -- seeded generator of whole `topsicle` command lines with their input files, for the CLI-level
differential between the reference's own Topsicle/main.py:main() (imported in the build container through oracle/ref_import.py)
and topsicle_amd.main, and for the fixtures oracle/gen_golden.py derives from it (tests/golden/cli_*.json).

A case = {"name", "files": {relative path: text or gz-of-text}, "pre": {relative path under the output directory: text},
"argv": [flags without -i / -o], "input": relative path (a file, or a directory), "exit": expected SystemExit code or None}.
What is compared (normalise()): telolengths_all.csv as a row SEQUENCE, the summary lines of the log (after the last "finished
processing all reads"), and every filtered file by name and content (Topsicle/main.py:52-154, 156-309).
"""
import gzip
import io
import os

import numpy as np

MOTIFS = ["CCCTAA", "CCCTAAA", "AAACCCT", "TTAGGG", "TTTAGGG"]


def _mutate(rng, s, sub, ins, dele):
    out = []
    for ch in s:
        r = rng.random()
        if r < dele:
            continue
        if r < dele + sub:
            out.append("ACGT"[rng.integers(0, 4)])
        else:
            out.append(ch)
        if rng.random() < ins:
            out.append("ACGT"[rng.integers(0, 4)])
    return "".join(out)


def _revcomp(s):
    return s[::-1].translate(str.maketrans("ACGTacgtNn", "TGCAtgcaNn"))


def make_read(rng, motif, length, telomeric, noisy):
    """A read of about `length` bases: a telomere tract (exact repeats from a random phase) followed by random sequence, as it is
    or reverse-complemented; a few lower-case stretches and N's (the reference upper-cases and N never matches)."""
    if telomeric:
        t = int(rng.integers(max(300, length // 6), max(400, length // 2)))
        ph = int(rng.integers(0, len(motif)))
        tract = (motif * (t // len(motif) + 2))[ph:ph + t]
    else:
        tract = ""
    rest = "".join("ACGT"[c] for c in rng.integers(0, 4, max(length - len(tract), 0)))
    s = tract + rest
    if noisy:
        s = _mutate(rng, s, 0.02, 0.01, 0.01)
    if rng.random() < 0.3:
        a = int(rng.integers(0, max(len(s) - 50, 1)))
        s = s[:a] + s[a:a + 40].lower() + s[a + 40:]
    if rng.random() < 0.2:
        a = int(rng.integers(0, max(len(s) - 5, 1)))
        s = s[:a] + "N" + s[a + 1:]
    if rng.random() < 0.5:
        s = _revcomp(s)
    return s


def file_text(rng, recs, fmt, wrap=None):
    out = io.StringIO()
    for rid, desc, s in recs:
        head = rid + (" " + desc if desc else "")
        if fmt == "fastq":
            q = "".join(chr(33 + int(c)) for c in rng.integers(2, 40, len(s)))
            out.write(f"@{head}\n{s}\n+\n{q}\n")
        elif wrap:
            out.write(f">{head}\n")
            for i in range(0, len(s), wrap):
                out.write(s[i:i + wrap] + "\n")
        else:
            out.write(f">{head}\n{s}\n")
    return out.getvalue()


def make_case(seed):
    rng = np.random.default_rng(seed)
    motif = MOTIFS[int(rng.integers(0, len(MOTIFS)))]
    n_files = int(rng.integers(1, 4))
    window = int(rng.choice([100, 100, 60, 150]))
    trim = int(rng.choice([100, 100, 0, 37]))
    min_len = int(rng.choice([1200, 1500, 2000]))
    maxlen = int(rng.choice([20000, 2500, 1800]))
    files, all_ids = {}, []
    for f in range(n_files):
        n_reads = int(rng.integers(4, 10))
        recs = []
        for i in range(n_reads):
            length = int(rng.choice([min_len - 200, min_len, min_len + 1, maxlen - 1 if maxlen < 5000 else 2600, 2200, 3000, 3500]))
            length = max(length, 700)
            telomeric = rng.random() < 0.7
            rid = f"r{seed}_{f}_{i}"
            recs.append((rid, "len=%d" % length if rng.random() < 0.5 else "", make_read(rng, motif, length, telomeric, rng.random() < 0.6)))
            if telomeric and length > min_len + 300:
                all_ids.append(rid)
        kind = ["fastq", "fastq.gz", "fasta", "fa.gz", "fq"][int(rng.integers(0, 5))]
        fmt = "fastq" if kind.startswith("f") and "q" in kind.split(".")[0] else "fasta"
        text = file_text(rng, recs, fmt, wrap=int(rng.choice([0, 60, 80])) if fmt == "fasta" else None)
        files[f"in/reads_{f}.{kind}"] = text
    argv = ["--pattern", motif, "--minSeqLength", str(min_len), "--threads", "1"]
    if rng.random() < 0.6:
        ks = sorted({int(k) for k in rng.integers(4, len(motif) + 1, int(rng.integers(1, 4)))}, reverse=bool(rng.random() < 0.3))
        argv += ["--telophrase"] + [str(k) for k in ks]
    if rng.random() < 0.6:
        cs = [float(c) for c in rng.choice([0.3, 0.4, 0.5, 0.6, 0.7, 0.8], int(rng.integers(1, 4)), replace=False)]
        argv += ["--cutoff"] + [str(c) for c in cs]
    if rng.random() < 0.6:
        argv += ["--slide", str(int(rng.choice([6, 7, 5, 10, 3])))]
    if window != 100:
        argv += ["--windowSize", str(window)]
    if trim != 100:
        argv += ["--trimfirst", str(trim)]
    if maxlen != 20000:
        argv += ["--maxlengthtelo", str(maxlen)]
    pre, exit_code = {}, None
    r = rng.random()
    if r < 0.12:
        pre["telolengths_all.csv"] = "file_number,phrase,trc,readID,telo_length\nold,5,0.900,x,100\n"
        if rng.random() < 0.5:
            argv += ["--override"]
        else:
            exit_code = 1
    single = n_files == 1
    if single and all_ids and rng.random() < 0.6:
        argv += ["--read_check", all_ids[int(rng.integers(0, len(all_ids)))]]
    # --rawcountpattern (BASELINE configs[4]'s flag; Topsicle/main.py:146-150 -> rawcount_{k}_{i}.csv per read).  Drawn from a generator of
    # its own so that the cases of the seeds recorded earlier keep their inputs; never with --read_check (upstream's branch names an
    # undefined variable there: main.py:118)
    if "--read_check" not in argv and np.random.default_rng(seed ^ 0x7A3C).random() < 0.35:
        argv += ["--rawcountpattern"]
    return {"name": f"case{seed}", "files": files, "pre": pre, "argv": argv, "input": next(iter(files)) if single else "in", "exit": exit_code}


def materialise(case, root):
    """Write the case's input files (and pre-existing outputs) under `root`; returns (input path, output dir)."""
    for rel, text in case["files"].items():
        p = os.path.join(root, rel)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        if rel.endswith(".gz"):
            with gzip.GzipFile(p, "wb", mtime=0) as h:
                h.write(text.encode())
        else:
            with open(p, "w") as h:
                h.write(text)
    out = os.path.join(root, "out")
    os.makedirs(out, exist_ok=True)
    for rel, text in case.get("pre", {}).items():
        with open(os.path.join(out, rel), "w") as h:
            h.write(text)
    return os.path.join(root, case["input"]), out


SUMMARY_STARTS = ("k-mer:", "asymptotic TRC", "Median telomere length", "Asymptotic TRC", "Using median", "Using 0.9", "Quadratic fit",
                  "Maximum TRC", "No read has", "Not enough data", "All telomere found")


def normalise(out_dir):
    """What a run left behind, in comparable form."""
    import csv
    res = {"csv": None, "summary": [], "filtered": {}, "rawcount": {}}
    p = os.path.join(out_dir, "telolengths_all.csv")
    if os.path.exists(p):
        res["csv"] = [r for r in csv.reader(open(p, newline=""))]
    log = os.path.join(out_dir, "topsicle_run.log")
    if os.path.exists(log):
        lines = [ln.rstrip("\n").split("] ", 1)[1] if "] " in ln else ln.rstrip("\n") for ln in open(log)]
        last = max([i for i, ln in enumerate(lines) if ln.startswith("finished processing all reads")], default=-1)
        res["summary"] = [ln for ln in lines[last + 1:] if ln.startswith(SUMMARY_STARTS)]
    for f in sorted(os.listdir(out_dir)):
        if "_trc_over_" in f:
            res["filtered"][f] = open(os.path.join(out_dir, f)).read()
        if f.startswith("rawcount_") and f.endswith(".csv"):
            res["rawcount"][f] = open(os.path.join(out_dir, f)).read()
    return res
