#!/usr/bin/env python3
"""A/B builds of the default scan kernels: recompiles kernel groups 0 and 1 (plain and pair-table kernels) of
libtopsicle_hip.so with extra compiler flags and links them with the product build's other groups.

    python scripts/build_variant.py NAME [-DFLAG=..] ...   ->  topsicle_amd/libtopsicle_hip_NAME.so

Run a script against it with TOPSICLE_HIP_LIB=topsicle_amd/libtopsicle_hip_NAME.so (hiplib.load_library).
`--groups 0,1,3` recompiles other groups as well (tps_kernels.h lists them)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def main():
    argv = sys.argv[1:]
    groups = [0, 1]
    if "--groups" in argv:
        i = argv.index("--groups")
        groups = [int(x) for x in argv[i + 1].split(",")]
        del argv[i:i + 2]
    name, extra = argv[0], argv[1:]
    ge.build_hip()                                   # the product objects the variant links against
    base = os.path.join(ge.CSRC, "_build", "libtopsicle_hip")
    bdir = os.path.join(ge.CSRC, "_build", "variant_" + name)
    os.makedirs(bdir, exist_ok=True)
    jobs = []
    for g in groups:
        src = "topsicle_hip.hip" if g == 0 else "tps_kernels.hip"
        obj = os.path.join(bdir, f"group{g}.o")
        jobs.append((g, obj, subprocess.Popen(["hipcc"] + ge.HIP_FLAGS + extra + [f"-DTPS_KGROUP={g}", "-c", "-o", obj, os.path.join(ge.CSRC, src)], cwd=ge.CSRC)))
    objs = {g: os.path.join(base, f"group{g}.o") for g in range(ge.KGROUPS)}
    for g, obj, p in jobs:
        if p.wait() != 0:
            raise SystemExit(f"hipcc failed for group {g}")
        objs[g] = obj
    out = os.path.join(ROOT, "topsicle_amd", f"libtopsicle_hip_{name}.so")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + [objs[g] for g in range(ge.KGROUPS)], cwd=ge.CSRC)
    print(out)


if __name__ == "__main__":
    main()
