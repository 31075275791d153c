import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_present():
    # cheap probe that does not initialise any runtime: the KFD node only exists with a GPU
    return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    if _gpu_present():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _io_options_back_to_defaults():
    """Tests steer the native reader through seqio.io_option (tps_io_set_option: process-wide); every test starts from the defaults."""
    yield
    from topsicle_amd import seqio
    if seqio._io_lib:
        for key, val in seqio.IO_OPTION_DEFAULTS.items():
            seqio.io_option(key, val)


@pytest.fixture(scope="session")
def gold_dir():
    return GOLD


@pytest.fixture(scope="session")
def synth_cases():
    meta = json.load(open(os.path.join(GOLD, "synth_cases.json")))
    arrs = np.load(os.path.join(GOLD, "synth_cases.npz"))
    return meta, arrs


@pytest.fixture(scope="session")
def demo_windows():
    meta = json.load(open(os.path.join(GOLD, "demo_windows.json")))
    arrs = np.load(os.path.join(GOLD, "demo_windows.npz"))
    return meta, arrs


@pytest.fixture(scope="session")
def demo_records():
    """(id, seq) of the 44 demo reads, parsed by the test-side mini reader (not the product)."""
    import gzip
    recs = []
    with gzip.open(os.path.join(GOLD, "demo_col0.fastq.gz"), "rt") as h:
        while True:
            head = h.readline()
            if not head:
                break
            seq = h.readline().strip()
            h.readline()
            h.readline()
            recs.append((head[1:].split()[0], seq))
    return recs


@pytest.fixture
def emu_engine_factory():
    """() -> two engines with HipScanner's interface backed by the host emulation of the kernels (tests/emu): the host-side
    pipeline (batch.EnginePool) without a GPU."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import emu_driver
    emu_driver.build()
    from emu_engine import EmuEngine
    return lambda: [EmuEngine(), EmuEngine()]
