import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# every test run CHECKS the CFG-duplicate tag it relies on (unet3d.forward): a stale tag fails a test instead of producing a
# wrong latent
os.environ.setdefault("VDX_VERIFY_CFG_DUP", "1")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU; none visible")
    return torch.device("cuda:0")
