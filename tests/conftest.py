import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def built_lib():
    """libpwv_hip.so built in-tree (hipcc cross-compiles gfx950 without a GPU)."""
    from pwv_amd import _lib
    _lib.build_library()
    return _lib.lib()


@pytest.fixture()
def gpu(built_lib):
    import torch
    if not torch.cuda.is_available():
        pytest.fail('-m gpu tests need a GPU; none visible')
    return torch.device('cuda', 0)
