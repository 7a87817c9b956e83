"""Round 4's single-draw null (tests/util.py::psi_null_rule): where its per-gene summaries live and how they are read.
TEST INFRASTRUCTURE: imported by tests/test_gpu_fullsize.py and profiles/psi_null.py (which computes them)."""
import os

import numpy as np

from tests.support.psi_cases import ROOT

NULL_DIR = os.path.join(ROOT, "profiles", "psi_null")


def load_summary(path):
    z = np.load(path)
    s = {k: z[k] for k in ("shift", "n_gt", "max", "hist")}
    s["Nc"] = int(z["Nc"])
    return s
