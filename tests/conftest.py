import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a ROCm GPU (run with -m gpu on an MI355X box)")
    # a kernel plugin that is refused, cannot be built or cannot be checked is a RED test, not a warning inside a green one
    # (VERDICT r05 item 1c); the one test that provokes the refusal catches the warning itself (pytest.warns)
    config.addinivalue_line("filterwarnings", "error::beacon_amd.jit.JitWarning")


def golden(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


@pytest.fixture(scope="session")
def gold():
    return golden


def ref_to_dev(a):
    """reference [.., nx+2, ny+2] -> device layout [.., ny+2, nx+2]"""
    return np.ascontiguousarray(np.swapaxes(a, -1, -2))
