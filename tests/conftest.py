import os
import sys

import pytest

# No MIOpen / HIP environment overrides here: tests, bench.py and rpeflow_amd.evaluate all run the library's default
# convolution solvers (one policy, rpeflow_amd/runtime.py), and the parity bounds below are measured under them.

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
