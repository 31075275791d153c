"""Offline sweep of the CLI differential against the reference's own main() (build container only: needs /root/reference): the seeds
of tests/test_ref_cli_differential.py and any range beyond them, on the emulated engines.  Prints one line per seed that differs or that
the reference itself crashes on, and a summary.
usage: python3 scripts/ref_differential_sweep.py FIRST LAST"""
import os, sys, tempfile, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import cli_cases, ref_import                      # noqa: E402
import emu_engine                                 # noqa: E402
import test_ref_cli_differential as T             # noqa: E402

first, last = int(sys.argv[1]), int(sys.argv[2])
bad, crashed, ok, raw = [], [], 0, 0
for seed in range(first, last + 1):
    case = cli_cases.make_case(seed)
    with tempfile.TemporaryDirectory() as d:
        inp_r, out_r = cli_cases.materialise(case, os.path.join(d, "ref"))
        inp_p, out_p = cli_cases.materialise(case, os.path.join(d, "prod"))
        try:
            code_r = ref_import.run_reference_main(["-i", inp_r, "-o", out_r] + case["argv"])
        except Exception as e:                    # the reference's own crashes (IndexError on a reused .fasta, ...)
            crashed.append((seed, repr(e)[:100]))
            continue
        try:
            code_p = T.run_product([emu_engine.EmuEngine(), emu_engine.EmuEngine()], ["-i", inp_p, "-o", out_p] + case["argv"])
            assert code_r == code_p == case["exit"], ("exit codes", code_r, code_p, case["exit"])
            a, b = cli_cases.normalise(out_r), cli_cases.normalise(out_p)
            T.compare(a, b, (seed, case["argv"]))
            ok += 1
            raw += len(a["rawcount"])
        except Exception as e:
            bad.append(seed)
            print("DIFFERS seed", seed, case["argv"], repr(e)[:300], flush=True)
print(f"seeds {first}..{last}: {ok} equal ({raw} raw-count files among them), {len(bad)} different {bad}, reference crashed on {len(crashed)}: {crashed}")
