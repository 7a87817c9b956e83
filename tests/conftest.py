import os
import sys

import pytest

# The C oracle (oracle/brie_oracle.c) is OpenMP code that the GPU tests call between device calls: its idle worker threads
# must sleep, not spin, or they compete with the HIP runtime's own threads for the few host cores a GPU box grants
# (profiles/history/run_r3u.sh: 900 s against 50 s for the same work).  Set before libgomp is loaded.
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

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
