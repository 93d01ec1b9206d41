import os
import sys

import pytest

# Parity runs compare against the reference's CPU path (direct fp32 convolutions).  MIOpen's default fp32
# choice for the 3x3 convolutions of the 2-D branch is Winograd F(2,3)/F(3,2), whose transform arithmetic
# is ~1e-4 relative off a direct convolution -- enough to move EPE2D by 1e-3.  These knobs only affect
# MIOpen's solver choice for the dense convolutions, which are outside the hot path; nothing in
# rpeflow_amd reads them.  Must be set before the first convolution of the process.
os.environ.setdefault("MIOPEN_DEBUG_CONV_WINOGRAD", "0")
os.environ.setdefault("MIOPEN_DEBUG_CONV_FFT", "0")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
