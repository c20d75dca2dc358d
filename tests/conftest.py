import os
import sys

import pytest

try:  # torch bundles its own HIP runtime: it must be loaded before libfqzhip.so so both share one copy
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def sample_fq():
    with open(os.path.join(os.path.dirname(__file__), "golden", "sample.fq"), "rb") as f:
        return f.read()
