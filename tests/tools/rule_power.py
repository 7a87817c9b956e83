#!/usr/bin/env python
"""NEGATIVE CONTROLS of the parity rules (round 6): can the rules the HIP path is held to FAIL?

Every mutant is the reference's algorithm with ONE deliberate error, evaluated on the CPU in the position a wrong HIP
kernel would be in: "some other fp32 evaluation" (the o32b build of oracle/brie_oracle.c with the knobs of ensemble member
t6: float Box-Muller, cells cut into 6 parts walked in reverse, 64-cell fp32 partial sums, fused multiply-adds) PLUS the
error.  The mutant `none` is therefore member t6 itself -- the pipeline's own control: its summaries must equal the
committed ones bit for bit.  Each mutant is judged, constants untouched, by

  short   tests/util.py::states_close_violations (= assert_states_close of tests/test_gpu_parity.py) against the fp32
          oracle after 12 Adam steps on the SHORT shapes below (one stage of 12 steps at lr 0.01; six stages of 2 steps
          with the staged learning rates and a fresh optimiser each); one more shape sits on the +-9 clip and is held to
          the bounds of ::test_large_counts_and_clip (|Z_loc| <= 9, Z_loc within 5e-3 of the oracle's);
  fit     tests/util.py::psi_ensemble_rule against the COMMITTED member fixtures tests/golden/psi_ens_<case>_first64.npz
          after the case's whole default schedule (first 64 genes x all cells, the case's seeds) on FIT_CASES.

    python tests/tools/rule_power.py --short                       (seconds)
    python tests/tools/rule_power.py --fit --cores 6               (CPU, about two hours; profiles/_psi_cache/power/)
    python tests/tools/rule_power.py --freeze                      (tests/golden/rule_power*.{json,npz})

Reference semantics under test: model_TFProb.py:69,81 (clip), :234-241 (stages, fresh Adam), :194-211 (KL), :159 (one
draw per MC sample), :261-264 (loss_gene without kwargs), model_wrap.py:113-117 (pseudo-count on the two unique layers),
model_TFProb.py:176 (effLen columns 0, 4, 5), Keras Adam (epsilon 1e-7 outside the bias correction, 1 - beta in fp32).
The oracle is the checker here, never the thing measured.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from tests import util                                           # noqa: E402
from tests.support import ensemble as pe, psi_cases as pd        # noqa: E402

POWER = os.path.join(pd.CACHE, "power")
GOLDEN = pe.GOLDEN
TABLE = os.path.join(GOLDEN, "rule_power.json")
MEMBER = "t6"                                  # the ensemble member every mutant is built on
FIT_CASES = ("c2_cli_128", "c3_cli_128", "c2_api_512", "c1_kc0_cli_s8")
STATE = ("Z_loc", "Z_std_log", "Wc_loc", "intercept", "sigma_log")

# name: where the error is made ("c": oracle/brie_oracle.c -DBRIE_ORACLE_MUTANTS, enum MUT_*; "host": the driver below),
#       what it is, the reference line it violates, and where it cannot apply
MUTANTS = {
    "none": dict(kind="c", what="no error: ensemble member t6 itself (control)", ref="-"),
    "adam_eps_torch": dict(kind="c", what="Adam epsilon inside the bias correction (torch.optim.Adam's placement)", ref="Keras Adam; model_TFProb.py:237"),
    "adam_eps_1e8": dict(kind="c", what="Adam epsilon 1e-8 instead of Keras' 1e-7", ref="Keras Adam; model_TFProb.py:237"),
    "no_clip": dict(kind="c", what="no clip of Z_loc / intercept to [-9, 9]", ref="model_TFProb.py:69,81"),
    "beta2_double": dict(kind="c", what="1 - beta_2 formed in double (0.001) instead of fp32 (0.00100004673)", ref="Keras Adam"),
    "lr_rotated": dict(kind="host", what="the six stage learning rates rotated by one: .005 .001 .005 .01 .02 .01", ref="model_TFProb.py:234-241"),
    "kl_no_expm1": dict(kind="c", what="KL without 0.5 expm1(2 (rho - lambda)) and its gradients", ref="model_TFProb.py:208 (tfd.kl_divergence)"),
    "pseudo3": dict(kind="host", what="pseudo-count 0.01 on the third (ambiguous) layer too", ref="model_wrap.py:113-117", needs="L3"),
    "loss_gene_mc3": dict(kind="host", what="loss_gene drawn with MC_size 3 instead of 1", ref="model_TFProb.py:261-264", psi_rules=False),
    "no_moment_reset": dict(kind="host", what="ONE Adam over all six stages (moments and t not reset)", ref="model_TFProb.py:237"),
    "efflen_cols012": dict(kind="host", what="effLen columns 0, 1, 2 instead of 0, 4, 5", ref="model_TFProb.py:176", needs="L3"),
    "mc_same_noise": dict(kind="c", what="all MC samples of a step share sample 0's noise", ref="model_TFProb.py:159", needs="MC>1"),
    "no_bias_corr": dict(kind="c", what="Adam without bias correction (alpha = lr)", ref="Keras Adam"),
    "lik_grad_1pct": dict(kind="c", what="d loglik / dz too large by 1 %", ref="model_TFProb.py:162-185"),
    "lik_grad_01pct": dict(kind="c", what="d loglik / dz too large by 0.1 %", ref="model_TFProb.py:162-185"),
    "kl_grad_1pct": dict(kind="c", what="prior pull (mu - m) / sigma^2 on Z_loc too large by 1 %", ref="model_TFProb.py:208"),
    "sigma_grad_sign": dict(kind="c", what="sign of the s^2 / sigma^2 term in the sigma_log gradient flipped", ref="model_TFProb.py:208"),
}
LR_ROTATED = lambda lrs: list(lrs[-1:]) + list(lrs[:-1])        # noqa: E731

# the SHORT shapes: Nc, Ng, Kc, L, MC, data seed, model seed (+ "clip": counts and init that sit on the +-9 clip)
SHORT = {"l2_kc0_mc1": dict(Nc=300, Ng=64, Kc=0, L=2, MC=1, data_seed=5, seed=3),
         "l2_kc3_mc3": dict(Nc=257, Ng=100, Kc=3, L=2, MC=3, data_seed=6, seed=4),
         "l3_kc1_mc1": dict(Nc=400, Ng=60, Kc=1, L=3, MC=1, data_seed=7, seed=5),
         "l3_kc2_mc3": dict(Nc=300, Ng=64, Kc=2, L=3, MC=3, data_seed=8, seed=6),
         "clip": dict(Nc=64, Ng=32, Kc=0, L=2, MC=1, data_seed=9, seed=7, clip=True)}
SEQUENCES = {"one_stage_12": dict(stages=[(12, 0.01)], lr=0.01, fresh=1),
             "six_stages_2": dict(stages=None, lr=0.02, fresh=6)}       # (2, lr) for the six staged learning rates


def applies(mutant, L, MC):
    need = MUTANTS[mutant].get("needs")
    return not ((need == "L3" and L != 3) or (need == "MC>1" and MC <= 1))


def mutate_problem(P, mutant):
    """Host-side mutants that act on the inputs (returns counts_pc, effLen)."""
    counts, eff = [np.array(c, np.float32) for c in P["counts_pc"]], P["effLen"]
    if mutant == "pseudo3" and len(counts) > 2:
        idx = (P["counts"][0] + P["counts"][1]) > 0
        counts[2][idx] = counts[2][idx] + np.float32(0.01)
    if mutant == "efflen_cols012" and eff is not None:
        eff = np.array(eff, np.float32)
        eff[:, 4], eff[:, 5] = eff[:, 1].copy(), eff[:, 2].copy()
    return counts, eff


def run_schedule(o, stages, MC, mutant):
    if mutant == "lr_rotated":
        stages = list(zip([n for n, _ in stages], LR_ROTATED([lr for _, lr in stages])))
    for i, (n, lr) in enumerate(stages):
        if not (mutant == "no_moment_reset" and i > 0):
            o.reset_optimizer()
        o.minimize(n, lr, MC)


def make_oracle(P, seed, mutant, init=None):
    """`mutant` None: the fp32 oracle o32 (what HIP is compared with); else member t6's arithmetic + the mutant's error."""
    from oracle.c_oracle import COracle
    if mutant is None:
        return COracle(P["counts_pc"], P["Xc"], effLen=P["effLen"], seed=seed, init=init)
    counts, eff = mutate_problem(P, mutant)
    c_name = mutant if MUTANTS[mutant]["kind"] == "c" else "none"
    o = COracle(counts, P["Xc"], effLen=eff, seed=seed, mutant=c_name, init=init)
    m = pe.MEMBERS[MEMBER]
    o.set_parts(m["parts"])
    o.b_config(m["float_noise"], m["reverse"], m["chunk"])
    return o


def short_problem(name):
    from oracle.brie_oracle import OracleBRIE2
    s = SHORT[name]
    P = util.problem(s["Nc"], s["Ng"], s["Kc"], s["L"], seed=s["data_seed"])
    init = None
    if s.get("clip"):                          # deep one-sided coverage and a start 0.05 inside the clip: 12 steps at lr 0.01 cross it
        c1 = np.full((s["Nc"], s["Ng"]), 5000, np.float32)
        c1[:, ::2] = 0
        P = dict(P, counts=[c1, 5000 - c1], effLen=None)
        P["counts_pc"] = [c + np.float32(0.01) for c in P["counts"]]
        init = OracleBRIE2(s["Nc"], s["Ng"], 0, seed=s["seed"]).model_init()
        init["Z_loc"] = np.where(c1 > 0, 8.95, -8.95).astype(np.float32)
        init["intercept"] = np.where(c1[0] > 0, 8.95, -8.95).astype(np.float32)[None, :]
    return P, s, init


def run_short(mutants=None):
    """mutant -> shape -> sequence -> list of violations of the short-horizon rule against the fp32 oracle."""
    from oracle.brie_oracle import LEARNING_RATES
    out = {}
    for name in SHORT:
        P, s, init = short_problem(name)
        for seq, q in SEQUENCES.items():
            stages = q["stages"] or [(2, lr) for lr in LEARNING_RATES]
            base = make_oracle(P, s["seed"], None, init)
            run_schedule(base, stages, s["MC"], None)
            so = {k: np.array(getattr(base, k)) for k in STATE}
            for m in (mutants or MUTANTS):
                if MUTANTS[m].get("psi_rules") is False:
                    continue
                if not applies(m, s["L"], s["MC"]):
                    out.setdefault(m, {}).setdefault(name, {})[seq] = None
                    continue
                o = make_oracle(P, s["seed"], m, init)
                run_schedule(o, stages, s["MC"], m)
                sd = {k: np.array(getattr(o, k)) for k in STATE}
                if s.get("clip"):              # the bounds of tests/test_gpu_parity.py::test_large_counts_and_clip
                    v = ([("Z_loc", "beyond the clip", float(np.abs(sd["Z_loc"]).max()))] if np.abs(sd["Z_loc"]).max() > 9.0 else []) + \
                        ([("Z_loc", "atol 5e-3", float(np.abs(sd["Z_loc"] - so["Z_loc"]).max()))]
                         if not np.allclose(sd["Z_loc"], so["Z_loc"], atol=5e-3, rtol=0) else [])
                else:
                    v = util.states_close_violations(so, sd, lr=q["lr"], fresh=q["fresh"])
                out.setdefault(m, {}).setdefault(name, {})[seq] = [[str(x) if isinstance(x, str) else float(x) for x in t] for t in v]
    return out


def loss_gene_control():
    """`loss_gene_mc3` touches no state, so neither Psi rule can see it; what does is the accessor's own parity test
    (tests/test_gpu_parity.py::test_loss_gene_matches_oracle: rtol 1e-4, atol 1e-3 on the mean of 25 draws)."""
    Nc, Ng, Kc = 80, 100, 2
    P = util.problem(Nc, Ng, Kc, 2)
    res = {}
    for mc in (1, 3):
        o = util.oracle_model(P, Nc, Ng, Kc, 21, np.float64)
        o.minimize(P["counts_pc"], P["Xc"], 2, 0.01, 1)
        acc = np.zeros(Ng)
        for _ in range(25):
            acc += o.loss_and_grads(P["counts_pc"], P["Xc"], mc, need_grads=False)["loss_gene"]
        res[mc] = acc / 25
    err = np.abs(res[3] - res[1]) - (1e-3 + 1e-4 * np.abs(res[1]))
    return {"genes_outside_the_tolerance": int((err > 0).sum()), "of": Ng, "max_abs_difference": float(np.abs(res[3] - res[1]).max()),
            "rejected_by": "tests/test_gpu_parity.py::test_loss_gene_matches_oracle" if (err > 0).any() else None}


# ---- fit level ------------------------------------------------------------------------------------------------------------
def fit_path(case, mutant):
    return os.path.join(POWER, "%s_%s.npz" % (case, mutant))


def run_fit(case, mutant, threads):
    os.makedirs(POWER, exist_ok=True)
    out = fit_path(case, mutant)
    if os.path.exists(out):
        return
    P, c, n = pe.problem(case)
    t0 = time.time()
    o = make_oracle(P, pd.model_seed(pe.CASES[case]["of"]), mutant)
    o.set_threads(threads)
    run_schedule(o, pd.schedule(c["min_iter"]), c["MC"], mutant)
    np.savez(out, seconds=time.time() - t0, psi=np.asarray(o.Psi, np.float32), Wc_loc=np.asarray(o.Wc_loc, np.float64),
             intercept=np.asarray(o.intercept, np.float64), sigma_log=np.asarray(o.sigma_log, np.float64))
    print("%s %s: %.0f s on %d threads" % (case, mutant, time.time() - t0, threads), flush=True)


def fit_jobs(cases, mutants):
    jobs = []
    for case in cases:
        c = pd.CASES[pe.CASES[case]["of"]]
        for m in mutants:
            if MUTANTS[m].get("psi_rules") is False or not applies(m, c["L"], c["MC"]) or os.path.exists(fit_path(case, m)):
                continue
            jobs.append((case, m))
    jobs.sort(key=lambda j: -pd.CASES[pe.CASES[j[0]]["of"]]["Nc"] * pd.CASES[pe.CASES[j[0]]["of"]]["min_iter"])
    return jobs


def run_all_fits(cases, mutants, cores, threads=3):
    """Every missing (case, mutant) as its own process (the mutant id is a global of the loaded library)."""
    jobs, live = fit_jobs(cases, mutants), []
    while jobs or live:
        live = [p for p in live if p.poll() is None]
        while jobs and (len(live) + 1) * threads <= cores:
            case, m = jobs.pop(0)
            env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_WAIT_POLICY="active")
            live.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--one", "%s:%s:%d" % (case, m, threads)], env=env))
        time.sleep(2)


def summaries_of(case, mutant):
    psi_o32, par_o32, _ = pe.load_fixture(case)
    z = np.load(fit_path(case, mutant))
    return util.gene_summaries(z["psi"], psi_o32, pd.util_params({k: z[k] for k in pd.PARAMS}), par_o32)


def judge(case, s, members=None):
    if members is None:
        _, _, members = pe.load_fixture(case)
    rep = util.psi_ensemble_rule(s, members, "%s" % case, check=False)
    return {"rejected": not rep["holds"], "violated": [v[0] for v in rep.get("violated", [])],
            "stats": rep["hip_vs_o32"], "ensemble_max": {k: max(m[k] for m in rep["ensemble_vs_o32"].values())
                                                         for k in ("moved_genes", "quiet_rate", "quiet_p99", "undisplaced_max")}}


def freeze(cases):
    """tests/golden/rule_power_<case>.npz (every mutant's per-gene summaries against o32) + tests/golden/rule_power.json."""
    short = run_short()
    table = {"member": MEMBER, "mutants": {m: {k: v for k, v in d.items() if k != "kind"} for m, d in MUTANTS.items()},
             "short_shapes": SHORT, "short_sequences": {k: {kk: vv for kk, vv in v.items() if kk != "stages"} for k, v in SEQUENCES.items()},
             "fit_cases": list(cases), "short": {}, "fit": {}, "loss_gene_mc3": loss_gene_control(), "control": {}}
    for m, shapes in short.items():
        table["short"][m] = {"rejected": any(v for sq in shapes.values() for v in sq.values() if v),
                             "by_shape": {n: {q: (None if v is None else sorted(set("%s: %s" % (t[0], t[1]) for t in v)))
                                              for q, v in sq.items()} for n, sq in shapes.items()}}
    for case in cases:
        blob = {}
        _, _, members = pe.load_fixture(case)
        for m in MUTANTS:
            if not os.path.exists(fit_path(case, m)):
                continue
            s = summaries_of(case, m)
            for k in ("shift", "n_gt", "max", "hist"):
                blob["%s_%s" % (m, k)] = s[k]
            blob["Nc"] = s["Nc"]
            table["fit"].setdefault(m, {})[case] = dict(judge(case, s, members), seconds=round(float(np.load(fit_path(case, m))["seconds"]), 1))
            if m == "none":                                   # the control: member t6 of the committed fixture, bit for bit
                table["control"][case] = bool(all(np.array_equal(s[k], members[MEMBER][k]) for k in ("shift", "n_gt", "max", "hist")))
        if blob:
            np.savez_compressed(os.path.join(GOLDEN, "rule_power_%s.npz" % case), **blob)
    for m in MUTANTS:
        f = table["fit"].get(m, {})
        table["fit"].setdefault(m, {})
        table.setdefault("summary", {})[m] = {
            "short_horizon_rejects": table["short"].get(m, {}).get("rejected"),
            "ensemble_rejects_in": sorted(c for c, r in f.items() if r["rejected"]),
            "ensemble_accepts_in": sorted(c for c, r in f.items() if not r["rejected"])}
    with open(TABLE, "w") as fh:
        json.dump(table, fh, indent=1, sort_keys=True)
    print(json.dumps(table["summary"], indent=1))
    print("control (mutant `none` == member %s of the committed fixtures):" % MEMBER, table["control"])


def load_frozen(case):
    """mutant -> gene summaries of tests/golden/rule_power_<case>.npz."""
    z = np.load(os.path.join(GOLDEN, "rule_power_%s.npz" % case))
    out = {}
    for m in MUTANTS:
        if "%s_shift" % m in z.files:
            out[m] = dict({k: z["%s_%s" % (m, k)] for k in ("shift", "n_gt", "max", "hist")}, Nc=int(z["Nc"]))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--short", action="store_true")
    ap.add_argument("--fit", action="store_true")
    ap.add_argument("--freeze", action="store_true")
    ap.add_argument("--cases", default=",".join(FIT_CASES))
    ap.add_argument("--mutants", default=",".join(MUTANTS))
    ap.add_argument("--cores", type=int, default=6)
    ap.add_argument("--threads", type=int, default=3, help="OpenMP threads per fit (3 divides member t6's six parts)")
    ap.add_argument("--one", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.one:
        case, m, t = args.one.split(":")
        return run_fit(case, m, int(t))
    cases, mutants = [c for c in args.cases.split(",") if c], [m for m in args.mutants.split(",") if m]
    if args.short:
        for m, shapes in run_short(mutants).items():
            print(m, json.dumps({n: {q: (None if v is None else sorted(set("%s: %s" % (t[0], t[1]) for t in v))) for q, v in sq.items()}
                                 for n, sq in shapes.items()}))
        print("loss_gene_mc3", loss_gene_control())
    if args.fit:
        run_all_fits(cases, mutants, args.cores, args.threads)
    if args.freeze:
        freeze(cases)


if __name__ == "__main__":
    main()
