import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip():
    """The product library on cuda:0 -- fails loudly if the .so or the GPU is missing."""
    try:
        import torch      # noqa: F401  (before the library: two HIP runtimes in one process, the one loaded first owns the GPU;
    except ImportError:   #  tests that view canvas memory as torch tensors need torch's to be that one)
        pass
    import smrender_amd as sa
    sa.init(0)
    yield sa
    sa.shutdown()
