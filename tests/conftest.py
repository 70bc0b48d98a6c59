import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture
def deterministic_mode():
    """Fixed-order reductions in libasr_hip.so for the duration of a test (engines read the switch when they are built)."""
    from asr_chinese_e2e_amd import kernels
    old = kernels.set_deterministic(True)
    yield
    kernels.set_deterministic(old)
