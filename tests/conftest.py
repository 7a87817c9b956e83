import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lib():
    """The HIP extension; GPU tests must fail loudly (not skip) if it cannot be had.
    If the in-tree library is absent or older than its sources it is (re)built first (hipcc is in the image)."""
    from brie_amd import _capi
    from brie_amd.build import compile_library
    compile_library()
    return _capi.load_library()
