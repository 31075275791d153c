"""AddressSanitizer + UndefinedBehaviorSanitizer over the CPU-side native code: the FASTA/FASTQ reader + packer
(csrc/tps_io.cpp, csrc/tps_pack.h: mmap, zlib, thread team, AVX2 packer) and the host emulation of the kernel source
(tests/emu).  Both are rebuilt with -fsanitize=address,undefined and the reader / emulation tests are re-run against those
builds in a child interpreter (libasan has to be preloaded).  GPU AddressSanitizer is not available on this pool; the
device-only parts (DPP scans, LDS atomics) are covered by the -m gpu parity tests instead."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "emu", "_build")


def _runtime(name):
    p = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def _build_io_asan():
    src = os.path.join(ROOT, "topsicle_amd", "csrc", "tps_io.cpp")
    out = os.path.join(BUILD, "libtopsicle_io_asan.so")
    os.makedirs(BUILD, exist_ok=True)
    deps = [src, os.path.join(ROOT, "topsicle_amd", "csrc", "tps_pack.h"), os.path.join(ROOT, "topsicle_amd", "csrc", "tps_gzpar.h"),
            os.path.join(ROOT, "include", "topsicle_hip.h")]
    if not (os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps)):
        subprocess.check_call(["g++", "-O1", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-std=c++17", "-shared", "-fPIC",
                               "-Wall", "-o", out, src, "-lz", "-lpthread"])
    return out


def _run_under_sanitizers(pytest_args, extra_env):
    asan = _runtime("libasan.so")
    if asan is None:
        pytest.skip("no libasan in this toolchain")
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0:verify_asan_link_order=0",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", PYTHONMALLOC="malloc", **extra_env)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + pytest_args, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=1500)
    out = r.stdout + r.stderr
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-6000:]
    assert r.returncode == 0, out[-6000:]
    return out


def test_reader_and_packer_under_asan_ubsan():
    """Every reader test again (truncated / corrupt gzip and BGZF blocks, CRLF, wrapped records, records larger than the
    buffers, the packed mmap path and its hand-over to the streaming decoder) on the sanitizer build."""
    lib = _build_io_asan()
    out = _run_under_sanitizers(["tests/test_seqio_native.py"], {"TOPSICLE_IO_LIB": lib})
    assert " passed" in out and "skipped" not in out.split("passed")[-1]
    # round 5: byte-range readers (plain and BGZF), concurrent thread-team pools, the offset-based record writer
    # (the reader-level tests: the CLI-level ones import matplotlib, whose C++ exceptions a preloaded libasan cannot intercept)
    out = _run_under_sanitizers(["tests/test_shards.py", "-k", "seams or odd_records or bgzf_readers"], {"TOPSICLE_IO_LIB": lib})
    assert " passed" in out


def test_parallel_gzip_inflater_under_asan_ubsan():
    """csrc/tps_gzpar.h (speculative block starts, symbol buffers written through raw pointers, window markers) on every block
    type, on corrupt and truncated files: no out-of-bounds access whatever the input holds."""
    lib = _build_io_asan()
    out = _run_under_sanitizers(["tests/test_gzpar.py", "-k", "block_types or members or corrupt or fuzz or short_period"], {"TOPSICLE_IO_LIB": lib})
    assert " passed" in out


def test_kernel_emulation_under_asan_ubsan():
    """The kernel source compiled as the host emulation, with every LDS / global index checked: goldens, random reads,
    multi-tile reads, per-pattern tiles with chains, wide windows."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import emu_driver
    emu_driver.build(asan=True)
    out = _run_under_sanitizers(["tests/test_emulation.py", "-k",
                                 "synthetic_goldens or random_vs_oracle or multi_tile or per_pattern_tiles or wide_windows or demo_reads or "
                                 "clean_batch_layout or chains_of_every_length or hand_over"],
                                {"TPS_EMU_ASAN": "1"})
    assert " passed" in out


def test_two_pass_route_under_asan_ubsan():
    """The heads-mode reader (tps_reader_next_heads), the second pass's packer (tps_pack_spans: spans into mapped / inflated text,
    wrapped FASTA joined on the fly) and the emulation's scan of both passes, on the sanitizer builds."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import emu_driver
    emu_driver.build(asan=True)
    lib = _build_io_asan()
    out = _run_under_sanitizers(["tests/test_two_pass.py"], {"TPS_EMU_ASAN": "1", "TOPSICLE_IO_LIB": lib})
    assert " passed" in out
