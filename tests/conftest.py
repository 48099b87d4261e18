import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def kat():
    with open(os.path.join(ROOT, "tests", "golden", "reference_kat.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session", autouse=True)
def _torch_before_the_library():
    """PyTorch-ROCm bundles its own HIP runtime (same soname as /opt/rocm's, which libaudiosync_hip.so links).
    Whichever is loaded first serves both; torch only sees the GPU when it is torch's, so the tests that use torch
    for device memory import it before the library is dlopen'ed (no effect on a box without a GPU)."""
    try:
        import torch
        torch.cuda.is_available()
    except Exception:
        pass
    yield
