"""Soak of the many-steps-per-launch path: tests/test_gpu_parity.py::test_many_steps_per_launch_are_bit_identical_to_the_two_launch_path
on RANDOM small shapes with fresh seeds (the committed family has 26):  python tests/tools/soak_fused_steps.py [n] [seed]
Exit code = number of failing cases."""
import os
import sys
import traceback

os.environ.setdefault("OMP_WAIT_POLICY", "passive")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np                                   # noqa: E402
from tests import test_gpu_parity as T               # noqa: E402
from brie_amd import _capi                           # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 717171
rng = np.random.default_rng(seed)
lib = _capi.load_library()
fails = 0
for i in range(n):
    Nc = int(rng.choice([1, 2, 15, 16, 17, 33, 64, 100, 200, 255, 256, 257, 300, 515, 700]))
    Ng = int(rng.choice([1, 3, 4, 5, 255, 256, 257, 500, 511, 700, 1025, 2000]))
    L = int(rng.choice([2, 3]))
    case = (10000 + i, Nc, Ng, int(rng.integers(0, 9)), L, int(rng.choice([1, 3])), bool(L == 3 or rng.random() < 0.3),
            bool(rng.random() < 0.15))
    try:
        T.test_many_steps_per_launch_are_bit_identical_to_the_two_launch_path(lib, *case)
    except Exception:
        fails += 1
        print("FAIL", case)
        traceback.print_exc()
print("fused-steps soak: %d random shapes (seed %d), failures: %d" % (n, seed, fails))
sys.exit(min(fails, 100))
