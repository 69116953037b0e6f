import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(GOLDEN, "reference_tests.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def live_torch():
    return np.load(os.path.join(GOLDEN, "live_torch.npz"))


@pytest.fixture(scope="session")
def live_fused():
    return np.load(os.path.join(GOLDEN, "live_fused_game.npz"))


@pytest.fixture(scope="session")
def live_list():
    return np.load(os.path.join(GOLDEN, "live_list.npz"))
