import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# the kernel-variant knobs that tests flip with monkeypatch.setenv between two launches are re-read per call only in this mode
# (csrc/common.hpp CREID_KNOB_ENV); must be in the environment before libcreid_hip.so makes its first launch
os.environ.setdefault("CREID_DEBUG_KNOBS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


@pytest.fixture(scope="session")
def golden():
    return load_golden
