"""CLI-level differential against the REFERENCE'S OWN main(): Topsicle/main.py is imported unchanged in the
build container (oracle/ref_import.py: Bio / seaborn / ruptures stand-ins, a functional SeqIO.write) and run on seeded random
inputs and flag sets (oracle/cli_cases.py: 1 - 3 files, fasta / fastq / gz, reads around --minSeqLength and --maxlengthtelo,
--telophrase / --cutoff lists, --slide, --trimfirst, --windowSize, --read_check, --override, a pre-existing telolengths_all.csv);
topsicle_amd.main runs the same command on the emulated engines.  Compared: the CSV as a row SEQUENCE, the summary lines of the
log, the filtered files by name and content, the exit code.  Skips without /root/reference (the GPU box): there the same runs
are replayed from tests/golden/cli_*.json (test_cli_golden_cases here on the emulation, tests/test_gpu_parity.py on the GPU)."""
import hashlib
import json
import os

import pytest

import cli_cases
import ref_import
from topsicle_amd import main as cli

HAVE_REF = os.path.isdir(ref_import.REFERENCE_ROOT)


def run_product(engines, argv):
    args = cli.build_parser().parse_args(argv)
    cli.tprint.logfile = cli.get_log_path(args)
    try:
        cli.analysis_run(args, engines=engines)
    except SystemExit as e:
        return e.code if e.code is not None else 0
    return None


def compare(a, b, what):
    assert a["csv"] == b["csv"], (what, "csv rows")
    assert a["summary"] == b["summary"], (what, "summary lines")
    assert sorted(a["filtered"]) == sorted(b["filtered"]), (what, "filtered file names")
    for f in a["filtered"]:
        assert a["filtered"][f] == b["filtered"][f], (what, f)
    assert sorted(a.get("rawcount", {})) == sorted(b.get("rawcount", {})), (what, "raw-count file names")
    for f in a.get("rawcount", {}):
        assert a["rawcount"][f] == b["rawcount"][f], (what, f)


@pytest.mark.skipif(not HAVE_REF, reason="needs /root/reference (build container only)")
@pytest.mark.parametrize("first", [100, 112, 124, 136])
def test_cli_matches_the_reference_main(first, tmp_path, emu_engine_factory):
    """Twelve seeded cases per call.  The reference itself crashes on some inputs the generator can produce -- FASTA input with
    several k: the second k REUSES the first k's `.fasta` (main.py:64-66) and a read that passes only at the second k is not in it,
    so `bound_res[0]` raises IndexError (main.py:131) -- those cases are counted, not compared (the product completes them)."""
    crashed = []
    for seed in range(first, first + 12):
        case = cli_cases.make_case(seed)
        inp_r, out_r = cli_cases.materialise(case, str(tmp_path / f"ref{seed}"))
        inp_p, out_p = cli_cases.materialise(case, str(tmp_path / f"prod{seed}"))
        try:
            code_r = ref_import.run_reference_main(["-i", inp_r, "-o", out_r] + case["argv"])
        except IndexError as e:
            crashed.append((seed, repr(e)))
            continue
        code_p = run_product(emu_engine_factory(), ["-i", inp_p, "-o", out_p] + case["argv"])
        assert code_r == code_p == case["exit"], (seed, case["argv"], code_r, code_p)
        compare(cli_cases.normalise(out_r), cli_cases.normalise(out_p), (seed, case["argv"]))
    assert len(crashed) <= 3, crashed


@pytest.mark.skipif(not HAVE_REF, reason="needs /root/reference (build container only)")
def test_existing_fasta_temp_file_is_reused_like_upstream(tmp_path, emu_engine_factory):
    """main.py:64-66: a `<name>_trc_over_<cutoff>.fasta` that is already there is used, not rewritten -- also for FASTQ input."""
    case = cli_cases.make_case(131)
    case["argv"] = ["--pattern", case["argv"][1], "--minSeqLength", "1200", "--threads", "1", "--cutoff", "0.5"]
    rel = next(iter(case["files"]))
    name = os.path.splitext(os.path.basename(rel))[0]
    # the file the first run wrote = every record as FASTA (upstream's step 2 reads ITS records; the product scans the input)
    text = case["files"][rel]
    recs = [(r.description, str(r.seq)) for r in ref_import._parse(__import__("io").StringIO(text), "fastq" if text.startswith("@") else "fasta")]
    case["pre"] = {f"{name}_trc_over_0.5.fasta": "".join(f">{d}\n{s}\n" for d, s in recs)}
    case["files"] = {rel: text}
    case["input"] = rel
    inp_r, out_r = cli_cases.materialise(case, str(tmp_path / "ref"))
    inp_p, out_p = cli_cases.materialise(case, str(tmp_path / "prod"))
    assert ref_import.run_reference_main(["-i", inp_r, "-o", out_r] + case["argv"]) is None
    assert run_product(emu_engine_factory(), ["-i", inp_p, "-o", out_p] + case["argv"]) is None
    a, b = cli_cases.normalise(out_r), cli_cases.normalise(out_p)
    compare(a, b, "existing fasta")
    assert a["filtered"][f"{name}_trc_over_0.5.fasta"] == case["pre"][f"{name}_trc_over_0.5.fasta"]
    assert "Using existing file" in open(os.path.join(out_p, "topsicle_run.log")).read()


def compare_replayed(a, b, what):
    """A recorded run against a replay on ANOTHER machine: both CLIs list a directory with os.walk, whose order is the file system's
    (the fixtures were recorded in the build container; the GPU box's scratch directory orders the same three names differently).
    So: per (k, file) the row SEQUENCE must be equal, the k's must come in the recorded order, and the files in ONE order for every k
    -- everything else (summary lines, filtered files) is compared as it is."""
    def blocks(rows):
        out, order = {}, []
        for r in rows[1:]:
            key = (r[1], r[0])
            if key not in out:
                out[key] = []
                order.append(key)
            else:
                assert order[-1] == key, (what, "rows of one file and k are not contiguous", key)
            out[key].append(r)
        return out, order
    assert (a["csv"] is None) == (b["csv"] is None)
    if a["csv"] is not None:
        assert a["csv"][0] == b["csv"][0]
        ba, oa = blocks(a["csv"])
        bb, ob = blocks(b["csv"])
        assert ba == bb, (what, "csv rows per file and k")
        ks = lambda order: [k for i, (k, _f) in enumerate(order) if i == 0 or order[i - 1][0] != k]
        assert ks(oa) == ks(ob), (what, "order of the k-mer lengths")
        per_k = {}
        for k, f in ob:
            per_k.setdefault(k, []).append(f)
        full = max(per_k.values(), key=len) if per_k else []
        for k, fs in per_k.items():
            assert fs == [f for f in full if f in fs], (what, "file order differs between k-mer lengths", per_k)
    assert a["summary"] == b["summary"], (what, "summary lines")
    assert sorted(a["filtered"]) == sorted(b["filtered"]), (what, "filtered file names")
    for f in a["filtered"]:
        assert a["filtered"][f] == b["filtered"][f], (what, f)
    # rawcount_{k}_{i}.csv is numbered per input file (main.py:90, 150): runs over several files leave, under each name, the LAST file's
    # read -- walk order again -- so their contents are compared for one-file runs only (recorded as digests)
    assert sorted(a.get("rawcount", {})) == sorted(b.get("rawcount", {})), (what, "raw-count file names")
    if len({r[0] for r in (a["csv"] or [])[1:]}) <= 1:
        for f, d in a.get("rawcount", {}).items():
            t = b["rawcount"][f]
            assert d == {"sha256": hashlib.sha256(t.encode()).hexdigest(), "bytes": len(t)}, (what, f)


def golden_cases(gold_dir):
    return sorted(f for f in os.listdir(gold_dir) if f.startswith("cli_") and f.endswith(".json"))


def test_cli_golden_cases(gold_dir, tmp_path, emu_engine_factory):
    """The runs oracle/gen_golden.py recorded from the reference's main(): replayed on the emulated engines."""
    names = golden_cases(gold_dir)
    assert len(names) >= 5
    for n in names:
        g = json.load(open(os.path.join(gold_dir, n)))
        inp, out = cli_cases.materialise(g["case"], str(tmp_path / n))
        code = run_product(emu_engine_factory(), ["-i", inp, "-o", out] + g["case"]["argv"])
        assert code == g["case"]["exit"]
        compare_replayed(g["expected"], cli_cases.normalise(out), n)


@pytest.mark.gpu
def test_cli_golden_cases_on_gpu(gold_dir, tmp_path):
    """The same recorded runs of the reference's main() through the real CLI on the MI355X (fresh contexts, the native reader)."""
    names = golden_cases(gold_dir)
    assert len(names) >= 5
    for n in names:
        g = json.load(open(os.path.join(gold_dir, n)))
        inp, out = cli_cases.materialise(g["case"], str(tmp_path / n))
        try:
            cli.main(["-i", inp, "-o", out] + g["case"]["argv"])
            code = None
        except SystemExit as e:
            code = e.code if e.code is not None else 0
        assert code == g["case"]["exit"]
        compare_replayed(g["expected"], cli_cases.normalise(out), n)


def _dup_file(tmp_path):
    import numpy as np
    rng = np.random.default_rng(9)
    a = cli_cases.make_read(rng, "CCCTAA", 2600, True, False)
    b = cli_cases.make_read(rng, "CCCTAA", 1700, True, True)
    c = cli_cases.make_read(rng, "CCCTAA", 3100, True, True)
    fa = tmp_path / "dups.fasta"
    fa.write_text(f">dup first\n{b}\n>solo\n{a}\n>dup second\n{c}\n")
    return str(fa), {"b": b, "a": a, "c": c}


@pytest.mark.skipif(not HAVE_REF, reason="needs /root/reference (build container only)")
def test_duplicate_read_ids_per_read_api_like_upstream(tmp_path):
    """Two records with one id (malformed input).  Upstream's bound_detect has no `break` and overwrites maxlengthtelo with the
    first matching record's length when that is shorter (allsteps.py:257-264): the second record is clipped by the shortest
    record before it.  The per-read API reproduces that (6 of 6 fuzz differences were this)."""
    from emu_engine import EmuEngine
    from topsicle_amd import allsteps
    ref = ref_import.load_reference_allsteps()
    fa, _ = _dup_file(tmp_path)
    allsteps.set_engine(EmuEngine())
    try:
        for tail in ("forward", "reverse", None):
            for M in (20000, 2000, 1500):
                want = ref.bound_detect(fa, "dup", "CCCTAA", 100, 6, 100, M, 4, tail=tail)
                got = allsteps.bound_detect(fa, "dup", "CCCTAA", 100, 6, 100, M, 4, tail=tail)
                assert got == want, (tail, M)
                assert len(got) == (4 if tail is None else 2)
    finally:
        allsteps.set_engine(None)


def test_duplicate_read_ids_in_the_batched_cli_are_records_of_their_own(tmp_path, emu_engine_factory):
    """DELIBERATE DEVIATION: upstream's CLI looks every passing id up again in the filtered file and takes
    `bound_res[0]` -- with duplicate ids BOTH rows get the FIRST record's boundary, computed with the LAST record's tail
    (main.py:60, 125-133; allsteps.py:257-297).  The batched CLI reports every record by its own sequence and tail."""
    import csv
    import topsicle_oracle as orc
    fa, seqs = _dup_file(tmp_path)
    out = tmp_path / "o"
    assert run_product(emu_engine_factory(), ["-i", fa, "-o", str(out), "--pattern", "CCCTAA", "--minSeqLength", "1000", "--cutoff", "0.3"]) is None
    rows = list(csv.reader(open(out / "telolengths_all.csv")))[1:]
    assert [r[3] for r in rows] == ["dup", "solo", "dup"]
    pats = orc.kmer_table("CCCTAA", 4)
    for r, key in zip(rows, ("b", "a", "c")):
        cs, ce = orc.trc_counts(seqs[key], pats)
        call = orc.trc_call(cs, ce, pats, 6, 0.3)
        assert f"{call[2]:.3f}" == r[2]
        assert orc.step2(seqs[key], call[1], pats, 100, 6, 100, 20000) == int(r[4])
