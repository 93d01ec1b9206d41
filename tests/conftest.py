import os
import sys

import pytest

# One runtime policy for tests, bench.py and rpeflow_amd.evaluate (rpeflow_amd/runtime.py): the HIP graph-queue count it
# sets and MIOpen's default solvers; nothing else is overridden.  The parity bounds below are measured under it.

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from rpeflow_amd import runtime  # noqa: E402

runtime.configure()  # (before anything initialises the GPU)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
