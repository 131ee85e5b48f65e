import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Two tests drive the device through bench.py's harness, which keeps its buffers in torch tensors.  The PyTorch wheel
    # brings its own HIP runtime; it has to be the first one initialised in the process (as in bench.py, where torch is
    # imported before libmpcx.so is loaded) -- initialised after libmpcx's it finds no device.
    if os.path.exists("/dev/kfd") and "not gpu" not in (config.getoption("-m") or ""):
        try:
            import torch
            torch.cuda.is_available()
        except Exception:                                  # torch missing or broken: only those two tests will fail
            pass


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
