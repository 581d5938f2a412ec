import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _f32_products_exact_by_default():
    """the f32 product mode (exact, or "f32x3": three bf16 MFMAs) is a per-thread property of the calling engine (ops.split_products):
    every test starts from the exact f32 products, whatever engine the previous test left behind on this thread"""
    try:
        import torch
        if torch.cuda.is_available():
            from emoasr_amd import ops
            ops.split_products(False)
    except Exception:
        pass
    yield
