import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "fusion-method-for-video-frame-interpolation_amd")
for p in (ROOT, PKG_DIR):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # A gpu-marked test on a box without a GPU is an error in how the suite was invoked,
    # not a skip: the driver selects with -m gpu / -m "not gpu".
    pass


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def device():
    import torch
    assert torch.cuda.is_available(), "gpu-marked tests need a HIP device"
    return torch.device("cuda:0")
