"""NEGATIVE CONTROLS: do the parity rules the HIP path is held to REJECT a wrong algorithm?  (not gpu)

tests/tools/rule_power.py builds mutants of the CPU oracle -- the reference's algorithm with ONE deliberate error each, on top of
the arithmetic of ensemble member t6, i.e. in the position a wrong HIP kernel would be in -- and judges them with the two rules,
constants untouched: the short-horizon rule (tests/util.py::states_close_violations = tests/test_gpu_parity.py::
assert_states_close, 12 Adam steps) and the fit-level rule (tests/util.py::psi_ensemble_rule against the COMMITTED member
fixtures, whole default schedules).  The verdicts are committed (tests/golden/rule_power.json) together with every mutant's
per-gene summaries (tests/golden/rule_power_<case>.npz), so that this file can
  * re-run the short-horizon half live (seconds) and demand the committed verdicts,
  * re-judge the frozen fit-level summaries with the rule AS IT IS NOW and demand the committed verdicts (a loosened constant
    shows up here as a mutant that is no longer rejected),
  * re-run the smallest fit-level case live for three mutants,
  * state what the table says: which errors are rejected by which rule, and which by neither.
Reference semantics under test: model_TFProb.py:69,81,159,176,194-211,234-241,261-264; model_wrap.py:113-117; Keras Adam.
"""
import json
import os

import numpy as np
import pytest

from tests import util
from tests.support import ensemble as pe
from tests.tools import rule_power as rp

TABLE = json.load(open(rp.TABLE)) if os.path.exists(rp.TABLE) else None
pytestmark = pytest.mark.skipif(TABLE is None, reason="tests/golden/rule_power.json missing: python tests/tools/rule_power.py "
                                "--fit --freeze")

# the eight errors VERDICT r5 names (+ the controls this build added); `needs` narrows where an error can apply at all
REQUIRED = ("adam_eps_torch", "adam_eps_1e8", "no_clip", "beta2_double", "lr_rotated", "kl_no_expm1", "pseudo3", "loss_gene_mc3")
# errors whose effect is far above fp32 rounding: BOTH rules must reject them wherever they apply
GROSS = ("lr_rotated", "kl_no_expm1", "no_moment_reset", "mc_same_noise", "no_bias_corr", "lik_grad_1pct", "kl_grad_1pct",
         "sigma_grad_sign", "efflen_cols012")


def test_the_control_is_member_t6_bit_for_bit():
    """Mutant `none` = the mutant build with the switch off and member t6's knobs: the committed fixtures' t6 summaries must come
    back exactly (recorded at freeze time for every case), and 12 steps must equal the plain o32b build bit for bit now."""
    assert TABLE["control"] and all(TABLE["control"].values()), TABLE["control"]
    from oracle.c_oracle import COracle
    P = util.problem(120, 40, 2, 3, seed=3)
    m = pe.MEMBERS[rp.MEMBER]
    a = rp.make_oracle(P, 5, "none")
    b = COracle(P["counts_pc"], P["Xc"], effLen=P["effLen"], seed=5, variant_b=True)
    b.set_parts(m["parts"])
    b.b_config(m["float_noise"], m["reverse"], m["chunk"])
    for o in (a, b):
        o.minimize(12, 0.01, 3)
    for k in rp.STATE:
        assert np.array_equal(getattr(a, k), getattr(b, k)), k


def test_short_horizon_rule_live_matches_the_committed_verdicts():
    live = rp.run_short()
    for m, shapes in live.items():
        rejected = any(v for sq in shapes.values() for v in sq.values() if v)
        assert rejected == TABLE["short"][m]["rejected"], (m, shapes)
    assert not TABLE["short"]["none"]["rejected"]                     # the rule accepts a legitimate fp32 evaluation


@pytest.mark.parametrize("case", TABLE["fit_cases"] if TABLE else [])
def test_ensemble_rule_as_it_is_now_gives_the_committed_verdicts(case):
    """The frozen per-gene summaries of every mutant, judged by tests/util.py::psi_ensemble_rule against the committed member
    fixtures: the verdict must be the committed one.  Nobody may retune a constant without this table changing."""
    path = os.path.join(rp.GOLDEN, "rule_power_%s.npz" % case)
    if not os.path.exists(path):
        pytest.fail("%s missing: python tests/tools/rule_power.py --fit --freeze --cases %s" % (os.path.relpath(path, rp.ROOT), case))
    _, _, members = pe.load_fixture(case)
    frozen = rp.load_frozen(case)
    assert frozen, case
    for m, s in frozen.items():
        j = rp.judge(case, s, members)
        want = TABLE["fit"][m][case]
        assert j["rejected"] == want["rejected"] and j["violated"] == want["violated"], (case, m, j, want)
    assert not TABLE["fit"]["none"][case]["rejected"]


def test_ensemble_rule_live_on_the_smallest_case():
    """configs[0]'s shape (c1_kc0_cli_s8: 200 cells x 64 genes, 4 998 steps, MC_size 3) run NOW for the control, the smallest
    gradient error and one sub-rounding error: same verdicts as committed (the bits may differ between hosts: FMA or not)."""
    case = "c1_kc0_cli_s8"
    P, c, n = pe.problem(case)
    psi_o32, par_o32, members = pe.load_fixture(case)
    from tests.support import psi_cases as pd
    for m in ("none", "lik_grad_01pct", "adam_eps_1e8"):
        o = rp.make_oracle(P, pd.model_seed(pe.CASES[case]["of"]), m)
        o.set_threads(2)
        rp.run_schedule(o, pd.schedule(c["min_iter"]), c["MC"], m)
        s = util.gene_summaries(np.asarray(o.Psi, np.float32), psi_o32, util.run_params(o), par_o32)
        assert rp.judge(case, s, members)["rejected"] == TABLE["fit"][m][case]["rejected"], m


def test_what_the_table_says():
    """The statement DESIGN section 2 makes, held to the table: gross errors are rejected by BOTH rules wherever they apply;
    every error VERDICT r5 names is rejected by at least one test of the suite or is shown to sit below fp32 rounding (it moves
    no more genes than a legitimate second evaluation does); loss_gene_mc3 touches no state and is caught by the accessor's
    own parity test."""
    S, F = TABLE["short"], TABLE["fit"]
    for m in GROSS:
        assert S[m]["rejected"], m
        judged = {c: r for c, r in F[m].items()}
        assert all(r["rejected"] for r in judged.values()), (m, {c: r["rejected"] for c, r in judged.items()})
        assert judged or set(TABLE["fit_cases"]) != set(rp.FIT_CASES), m      # the complete table judges every one of them
    for m in REQUIRED:
        if m == "loss_gene_mc3":
            assert TABLE["loss_gene_mc3"]["rejected_by"] and TABLE["loss_gene_mc3"]["genes_outside_the_tolerance"] > 0
            continue
        caught = S[m]["rejected"] or any(r["rejected"] for r in F[m].values())
        if not caught:       # then it must be indistinguishable: inside the ensemble's own spread on every judged statistic
            for c, r in F[m].items():
                assert not r["violated"], (m, c)
        # recorded either way
        assert m in TABLE["summary"]
    # the smallest systematic error tried -- the likelihood gradient off by 0.1 % -- is rejected by both rules
    assert S["lik_grad_01pct"]["rejected"] and all(r["rejected"] for r in F["lik_grad_01pct"].values())
