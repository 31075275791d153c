"""Diagnostic (GPU box): the BGZF leg of topsicle_amd/e2e.py with the in-tree inflater and with zlib (TPS_IO_BGZF_ZLIB=1)."""
import json, os, subprocess, sys
code = ("import sys, json; sys.path.insert(0, '.'); from topsicle_amd import e2e, synth; "
        "b, o, t = synth.make_reads(10000, 15000, 'CCCTAA', seed=20250920, errors=synth.ONT); "
        "r = e2e.measure(b, o, 'CCCTAA', 4, 6, device=0, with_cli=False); "
        "print(json.dumps({k: r[k] for k in ('bgzf_noisy_quality_file_to_results', 'gz_noisy_quality_file_to_results')}))")
for mode in ("own", "zlib"):
    env = dict(os.environ)
    if mode == "zlib":
        env["TPS_IO_BGZF_ZLIB"] = "1"
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    print(mode, {k: (round(v["value"] / 1e9, 3), v["seconds_median"]) for k, v in d.items()})
