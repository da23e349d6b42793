import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests must never silently pass on a CPU-only box: they are deselected by `-m "not gpu"`; if someone
    # runs them without a GPU they fail loudly in the test body (no skip).
    pass


@pytest.fixture(scope="session")
def cuda():
    import torch
    assert torch.cuda.is_available(), "this test needs a GPU (marked @pytest.mark.gpu)"
    return torch.device("cuda:0")
