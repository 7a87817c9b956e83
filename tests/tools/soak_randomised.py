"""Soak run of the two seeded random test families of tests/test_gpu_parity.py with OTHER seeds than the committed
lists (the oracle is the checker, as in the tests):  python tests/tools/soak_randomised.py [n_shapes] [n_sequences] [seed] [n_wide_sequences] [n_wide_shapes]
Prints every failing case with the assertion message; exit code = number of failures."""
import os, sys, traceback
os.environ.setdefault("OMP_WAIT_POLICY", "passive")     # as tests/conftest.py: idle OpenMP workers of the C oracle sleep
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import test_gpu_parity as T
import json
import numpy as np

RECORD = os.environ.get("BRIE_SOAK_RECORD")     # path: collect what the cases NEED instead of asserting the state bounds
if RECORD:
    T._RECORD = []

n_shapes = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n_seq = int(sys.argv[2]) if len(sys.argv) > 2 else 120
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 424242
shape_fn = T.test_randomised_shapes_and_switches
seq_fn = T.test_randomised_operation_sequences
fails = 0
def wide_shapes(n, seed):
    """The shape / switch family over ARBITRARY shapes: any Nc in 1..520, any Ng in 1..1100, any Kc / Kg of the kind."""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        kind = ["plain", "plain", "wide", "cell", "xg", "fixed", "margin", "wide_cell", "wide_xg"][i % 9]
        L = int(rng.choice([2, 3]))
        Kc = int(rng.integers(9, 65)) if kind.startswith("wide") else int(rng.integers(0, 9))
        if kind.startswith("wide") and rng.random() < 0.25:
            Kc = int(rng.integers(65, 161))                  # very wide designs: 64-feature panels (round 4)
        Kg = int(rng.integers(1, 65)) if kind.endswith("xg") else 0
        if Kg and rng.random() < 0.25:
            Kg = int(rng.integers(65, 161))                  # very wide gene designs, in panels too
        out.append((5000 + i, kind, int(rng.integers(1, 521)), int(rng.integers(1, 1101)), Kc, Kg, L,
                    int(rng.choice([1, 2, 3, 5])), bool(L == 3 or rng.random() < 0.3)))
    return out


n_wide_shapes = int(sys.argv[5]) if len(sys.argv) > 5 else 0
for case in list(T._random_cases(n_shapes, seed=seed)) + wide_shapes(n_wide_shapes, seed + 3):
    try:
        shape_fn(None, *case)
    except Exception as e:      # noqa
        fails += 1
        print("SHAPE CASE FAILED", case, "->", " | ".join(l for l in str(e).splitlines() if l.strip())[:600] or traceback.format_exc()[-600:], flush=True)
print("shape cases done:", n_shapes, "+ wide", n_wide_shapes, "failures so far:", fails, flush=True)
def wide_sequences(n, seed):
    """Sequences over ARBITRARY shapes (the committed family draws from short lists): any Nc in 1..400, any Ng in 1..1300
    -- every residue mod 4 and mod 256 --, Kc 0..12."""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        ops = [str(rng.choice(["step", "step", "mask", "unmask", "reset", "loss_gene", "tiling", "window", "read"]))
               for _ in range(8)]
        out.append((1000 + i, int(rng.integers(1, 401)), int(rng.integers(1, 1301)), int(rng.integers(0, 13)),
                    int(rng.choice([2, 3])), bool(rng.random() < 0.5), bool(rng.random() < 0.5), tuple(ops)))
    return out


n_wide = int(sys.argv[4]) if len(sys.argv) > 4 else 0
sequences = list(T._op_sequences(n_seq, seed=seed + 1)) + wide_sequences(n_wide, seed + 2)
for case in sequences:
    try:
        seq_fn(None, *case)
    except Exception as e:      # noqa
        fails += 1
        print("SEQUENCE FAILED", case, "->", " | ".join(l for l in str(e).splitlines() if l.strip())[:600] or traceback.format_exc()[-600:], flush=True)
print("sequences done:", n_seq, "+ wide", n_wide, "total failures:", fails, flush=True)
if RECORD:
    rec = T._RECORD
    big = [r for r in rec if r[1] >= 1000]
    out = {"calls": len(rec), "arrays_with_1000_plus_elements": len(big),
           "p99.9_max_over_large_arrays": max(r[2] for r in big), "p99.9_quantiles_large": np.percentile([r[2] for r in big], [50, 99, 100]).tolist(),
           "max_quantiles": np.percentile([r[3] for r in rec], [50, 90, 99, 99.9, 100]).tolist(),
           "calls_with_an_element_beyond_1e-3": sum(1 for r in rec if r[4] > 0),
           "largest_count_beyond_1e-3": max(r[4] for r in rec),
           "worst_calls": sorted(rec, key=lambda r: -r[3])[:12]}
    with open(RECORD, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))
sys.exit(min(fails, 100))
