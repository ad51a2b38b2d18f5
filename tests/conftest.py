import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "sanitizer: the emulator suite under ASan + UBSan / TSan (slow; select with -m sanitizer or WSNARK_SANITIZERS=1)")


def pytest_collection_modifyitems(config, items):
    # the sanitizer runs take tens of minutes: only when asked for by name
    if "sanitizer" in (config.getoption("-m") or "") and "not sanitizer" not in (config.getoption("-m") or ""):
        return
    if os.environ.get("WSNARK_SANITIZERS") == "1":
        return
    skip = pytest.mark.skip(reason="sanitizer runs are selected with -m sanitizer (or WSNARK_SANITIZERS=1)")
    for it in items:
        if "sanitizer" in it.keywords:
            it.add_marker(skip)


def hamming_ok(b32):
    """The reference's own sanity check of drawn blinding (test/bn128_prover.js:65-71): the zeros among the significant bits
    of the 256-bit value number 96..160 (a uniform draw has 128 +- 8)."""
    z = bin(int.from_bytes(b32, "little"))[2:].count("0")
    return 96 <= z <= 160


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def orc():
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture
def tune():
    """tune(lib, NAME, value): a library switch for the duration of one test, through wsnark_tuning_set.  (The library reads the
    environment ONCE per switch name and process -- on first use -- so tests never rely on changing WSNARK_* variables mid-run.)"""
    done = []

    def set_(lib, name, value):
        lib.tune(name, value)
        done.append((lib, name))

    yield set_
    for lib, name in done:
        lib.tune(name, None)
