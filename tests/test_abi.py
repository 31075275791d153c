"""The C-ABI library loads on a machine without a GPU, exports every symbol include/*.h
declares, and refuses to create a context (there is no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from topsicle_amd import hiplib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "topsicle_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tps_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(hiplib.EXPORTS)


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(hiplib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert hiplib.load_library().tps_abi_version() == 4


def test_struct_layouts_match_header():
    assert C.sizeof(hiplib.Params) == 40
    assert hiplib.RESULT_DTYPE.itemsize == 48 and hiplib.RESULT_DTYPE.fields["flags"][1] == 40
    assert hiplib.RESULT_DTYPE.fields["gain"][1] == 32
    assert hiplib.DESC_DTYPE.itemsize == 16 and hiplib.DESC_DTYPE.fields["len"][1] == 8 and hiplib.DESC_DTYPE.fields["flags"][1] == 12


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="GPU present: context creation is expected to work")
def test_no_gpu_means_loud_failure():
    with pytest.raises(hiplib.TopsicleHipError) as e:
        hiplib.HipScanner(0)
    assert "no CPU fallback" in str(e.value)


def test_window_count_helper_matches_reference_formula():
    lib = hiplib.load_library()
    for L, W, s, t, M in [(15000, 100, 6, 100, 20000), (30000, 100, 6, 100, 20000), (199, 100, 6, 100, 20000),
                          (200, 100, 6, 100, 20000), (206, 100, 6, 100, 20000), (0, 100, 6, 100, 20000), (500, 37, 1, 0, 300)]:
        want = len(range(0, max(min(L, M) - t, 0) - W + 1, s)) if min(L, M) - t >= W else 0
        assert lib.tps_window_count(L, W, s, t, M) == want == hiplib.window_count(L, W, s, t, M)


def io_declared_symbols():
    text = open(os.path.join(ROOT, "include", "topsicle_io.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tps_[a-z0-9_]+)\s*\(", text)))


def test_io_header_binding_and_library_agree():
    """include/topsicle_io.h declares what libtopsicle_io.so exports and what seqio binds (half of the ingest
    boundary had no header)."""
    import subprocess
    from topsicle_amd import seqio
    assert io_declared_symbols() == sorted(seqio.IO_EXPORTS)
    lib = seqio._load_io()
    assert lib is not None
    for name in io_declared_symbols():
        assert hasattr(lib, name), name
    path = os.path.join(ROOT, "topsicle_amd", "libtopsicle_io.so")
    exported = {ln.split()[-1] for ln in subprocess.check_output(["nm", "-D", "--defined-only", path], text=True).splitlines() if " T " in ln}
    assert {s for s in exported if s.startswith("tps_")} == set(io_declared_symbols())          # nothing exported that the header does not declare
    assert lib.tps_io_set_option(b"no_such_option", 1) != 0 and b"unknown option" in lib.tps_io_last_error()


def test_hip_library_exports_nothing_undeclared_and_reads_no_environment():
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", hiplib.LIB_PATH], text=True).splitlines()
    exported = {ln.split()[-1] for ln in out if " T " in ln and ln.split()[-1].startswith("tps_") and
                not re.match(r"tps_(scan|binseg|followers|pack)_kernel", ln.split()[-1])}                # (the kernels' host stubs)
    assert exported == set(declared_symbols())
    for lib in (hiplib.LIB_PATH, os.path.join(ROOT, "topsicle_amd", "libtopsicle_io.so")):
        undefined = subprocess.check_output(["nm", "-D", "--undefined-only", lib], text=True)
        assert "getenv" not in undefined.replace("secure_getenv", ""), lib      # (the planner and the reader take their knobs through the ABI)
