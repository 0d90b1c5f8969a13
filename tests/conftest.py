"""pytest configuration: markers, paths, shared fixtures."""
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests need a real MI355X: on a box without one they are SKIPPED with the reason, not failed one by one
    (the product itself still fails loudly without a device, tests/test_abi.py)."""
    try:
        import torch
        have = torch.cuda.is_available()
    except Exception:
        have = False
    if have:
        return
    skip = pytest.mark.skip(reason="needs a HIP device (MI355X): run with gpurun / on the GPU box")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden


def bits(a):
    """Raw bytes view for bit-exact comparisons (distinguishes -0.0 from 0.0)."""
    return np.ascontiguousarray(a).view(np.uint8)
