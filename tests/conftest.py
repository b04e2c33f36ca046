import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """On a GPU box let torch bring up its HIP context before the library has run any kernel: torch's lazy initialisation
    reports "No HIP GPUs are available" when it comes after a long series of library handles in the same process (seen
    with the dist tests run last); the usual file order has them first, this makes any order work."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import binding
    return binding.lib()
