"""Soak run of the two seeded random test families of tests/test_gpu_parity.py with OTHER seeds than the committed
lists (the oracle is the checker, as in the tests):  python tests/tools/soak_randomised.py [n_shapes] [n_sequences] [seed]
Prints every failing case with the assertion message; exit code = number of failures."""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import test_gpu_parity as T

n_shapes = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n_seq = int(sys.argv[2]) if len(sys.argv) > 2 else 120
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 424242
shape_fn = T.test_randomised_shapes_and_switches
seq_fn = T.test_randomised_operation_sequences
fails = 0
for case in T._random_cases(n_shapes, seed=seed):
    try:
        shape_fn(None, *case)
    except Exception as e:      # noqa
        fails += 1
        print("SHAPE CASE FAILED", case, "->", str(e).splitlines()[0][:300] if str(e) else traceback.format_exc()[-400:], flush=True)
print("shape cases done:", n_shapes, "failures so far:", fails, flush=True)
for case in T._op_sequences(n_seq, seed=seed + 1):
    try:
        seq_fn(None, *case)
    except Exception as e:      # noqa
        fails += 1
        print("SEQUENCE FAILED", case, "->", str(e).splitlines()[0][:300] if str(e) else traceback.format_exc()[-400:], flush=True)
print("sequences done:", n_seq, "total failures:", fails, flush=True)
sys.exit(min(fails, 100))
