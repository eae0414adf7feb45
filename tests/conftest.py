import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# MIOpen's per-shape solver search (roma_amd.encoders.enable_miopen_find) costs ~30 s per fresh (process, set of conv shapes) and the
# suite builds models at many sizes: off by default here (the heuristic solver pick + planar VGG19 layout of round 2), switched on by
# the `miopen_find` fixture for the tests that pin the configuration bench.py times (560 -> 864: channels-last VGG19, searched solvers).
os.environ.setdefault("ROMA_MIOPEN_FIND", "0")


@pytest.fixture
def miopen_find():
    import torch
    old = torch.backends.cudnn.benchmark
    torch.backends.cudnn.benchmark = True
    try:
        yield
    finally:
        torch.backends.cudnn.benchmark = old


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
