import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(__file__))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    oracle.set_threads(1)
    return oracle


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import splatter_a_video_amd._lib as L
    L.lib()  # fails loudly when the HIP library was not built
    return torch.device("cuda:0")


@pytest.fixture
def lib_option():
    """set options of the library through its ABI (splat_set_option) for the duration of one test"""
    import splatter_a_video_amd._lib as L
    old = {}

    def set_(key, value):
        old.setdefault(key, L.get_option(key))
        L.set_option(key, value)

    yield set_
    for k, v in old.items():
        L.set_option(k, v)
