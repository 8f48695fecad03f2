import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `gpurun -- pytest -m gpu`)")


@pytest.fixture(scope="session")
def hip():
    """The loaded C-ABI library; GPU tests fail loudly (never skip) if it is missing."""
    import torch
    from selfocc_amd._lib import lib
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    return lib()
