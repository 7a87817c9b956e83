"""The pre-registered NULL ENSEMBLE of the fit-level parity rule (tests/util.py::psi_ensemble_rule): member definitions,
cases, and the loader of the committed fixtures tests/golden/psi_ens_<case>_first64.npz.  TEST INFRASTRUCTURE: imported by
tests/test_gpu_fullsize.py, tests/test_rule_power.py and profiles/psi_ensemble.py (which computes and freezes the members).
"""
import hashlib
import os

import numpy as np

from tests.support import psi_cases as pd

ROOT = pd.ROOT
GOLDEN = os.path.join(ROOT, "tests", "golden")
MANIFEST = os.path.join(GOLDEN, "psi_ensemble_manifest.json")
GENES = 64                                                  # the first 64 genes of a case (genes are independent)
# name: parts (the cells are cut into that many ranges with their own per-gene sums, as that many OpenMP threads would),
#       float Box-Muller (1) or the exact noise stream (0), the part's cells in reverse (1) or forward order,
#       cells per fp32 partial sum; all built with -ffp-contract=fast -mfma (oracle/c_oracle.py::build(variant_b=True))
MEMBERS = {"t2": dict(parts=2, float_noise=1, reverse=1, chunk=64),
           "t4": dict(parts=4, float_noise=1, reverse=1, chunk=64),
           "t6": dict(parts=6, float_noise=1, reverse=1, chunk=64),        # round 4's single draw
           "t8": dict(parts=8, float_noise=1, reverse=1, chunk=64),
           "t12": dict(parts=12, float_noise=1, reverse=1, chunk=64),
           "x3": dict(parts=3, float_noise=0, reverse=0, chunk=128)}       # sums / FMAs only: the noise stream is o32's
# round 4's cases (their 128-gene caches of t4 / t6 / t8 are sliced) and ONE new held-out set of seeds per shape
CASES = {"c2_cli_128": dict(of="c2_cli_128"), "c3_cli_128": dict(of="c3_cli_128"),
         "c2_cli_64_s5": dict(of="c2_cli_64_s5", held_out=True), "c3_cli_64_s5": dict(of="c3_cli_64_s5", held_out=True)}
# addendum 1 (registered after the four cases above had been judged, before anything ran on these): three more held-out cases
ADDENDUM_1 = {"c2_cli_64_s6": dict(of="c2_cli_64_s6", held_out=True), "c3_cli_64_s6": dict(of="c3_cli_64_s6", held_out=True),
              "mid_cli_64_s6": dict(of="mid_cli_64_s6", held_out=True)}
# addendum 2 (after addenda and cases above had been judged): the API schedule (BRIE2.fit defaults: 996 steps, MC_size 1) under
# the same rule -- round 4's three gene-sample cases (first 64 genes; o32 and t6 sliced from round 4's caches) and two held-out
ADDENDUM_2 = {"c2_api_512": dict(of="c2_api_512"), "c3_api_512": dict(of="c3_api_512"), "c3_api_512_s2": dict(of="c3_api_512_s2"),
              "c2_api_64_s7": dict(of="c2_api_64_s7", held_out=True), "c3_api_64_s7": dict(of="c3_api_64_s7", held_out=True)}
# addendum 3 (after everything above had been judged): the two BASELINE shapes no case had -- configs[4] and configs[0] -- held out
ADDENDUM_3 = {"c5_cli_64_s8": dict(of="c5_cli_64_s8", held_out=True), "c1_kc0_cli_s8": dict(of="c1_kc0_cli_s8", held_out=True)}
REGISTERED_FIRST = tuple(CASES)
CASES.update(ADDENDUM_1)
CASES.update(ADDENDUM_2)
CASES.update(ADDENDUM_3)
OLD_DRAWS = {"t4": "_t4", "t6": "", "t8": "_t8"}           # suffixes of profiles/_psi_cache/<case>_float32b<suffix>.npz
SLICED = ("c2_cli_128", "c3_cli_128", "c2_api_512", "c3_api_512", "c3_api_512_s2")      # cases with round-4 caches of more genes


def sha256(path):
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        for blk in iter(lambda: fh.read(1 << 22), b""):
            h.update(blk)
    return h.hexdigest()


def problem(case):
    """The case's problem restricted to its first GENES genes (exact for those genes: per-gene model, noise keyed by the
    global gene index)."""
    P, c = pd.problem(CASES[case]["of"])
    n = min(GENES, c["Ng"])
    P = dict(P, counts=[np.ascontiguousarray(x[:, :n]) for x in P["counts"]],
             counts_pc=[np.ascontiguousarray(x[:, :n]) for x in P["counts_pc"]],
             effLen=None if P["effLen"] is None else np.ascontiguousarray(P["effLen"][:n]))
    return P, c, n


def load_fixture(case):
    """(psi_o32, par_o32, members) of tests/golden/psi_ens_<case>_first64.npz -- what the GPU test consumes."""
    z = np.load(os.path.join(GOLDEN, "psi_ens_%s_first%d.npz" % (case, GENES)))
    zo = np.load(os.path.join(GOLDEN, str(z["o32_in"]))) if "o32_in" in z.files else z      # the o32 run: here or in round 4's fixture
    par = pd.util_params({k: zo[k] for k in pd.PARAMS})
    members = {m: dict({k: z["%s_%s" % (m, k)] for k in ("shift", "n_gt", "max", "hist")}, Nc=int(z["Nc"]))
               for m in MEMBERS if "%s_shift" % m in z.files}
    return zo["psi_o32"], par, members
