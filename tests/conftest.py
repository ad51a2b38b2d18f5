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
