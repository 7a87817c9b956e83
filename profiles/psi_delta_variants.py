#!/usr/bin/env python
"""PSI delta after the FULL default schedule for the model variants beyond the plain per-gene model:
gene features Xg with per-cell weights, per-cell intercept / sigma, wide cell designs, target="marginLik".
HIP vs the NumPy restatement (oracle/brie_oracle.py) in fp64, next to the same restatement in fp32 -- the reference's own
precision on the same trajectory.  In these models a sign event can hit a parameter shared by a whole ROW (a cell's Wg_loc
entry, its intercept) as well as a gene's own: exceedances are reported per column and per row.
    python profiles/psi_delta_variants.py [--out profiles/history/psi_delta_variants_r03.json]      (GPU box)
The oracle is the checker here, never the thing measured."""
import argparse
import json
import os
import sys
import time

import numpy as np

os.environ.setdefault("OMP_WAIT_POLICY", "passive")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

VARIANTS = {
    # name: mode, Kg, Kc, L, MC, target
    "gene_features_Kg2": ("gene", 2, 1, 2, 1, "ELBO"),
    "cell_intercept": ("cell", 0, 1, 2, 1, "ELBO"),
    "cell_intercept_Kg4_effLen_mc3": ("cell", 4, 2, 3, 3, "ELBO"),
    "gene_features_Kg9_lds_tile": ("gene", 9, 1, 2, 1, "ELBO"),
    "wide_Kc20_mfma_tile": ("gene", 0, 20, 2, 1, "ELBO"),
    "marginLik_mc3": ("gene", 0, 2, 2, 3, "marginLik"),
}
Nc, Ng = 200, 520
NULL_MEMBERS = (1, 2, 3, 4, 5)         # OracleBRIE2(variant_b=...)


def run_variant(name, min_iter=1000, seed=41, data_seed=37, null=False):
    """null=False (round 3): HIP and the fp32 NumPy oracle, each against the fp64 one.  null=True (round 4): HIP against
    the fp32 oracle, next to FIVE further fp32 NumPy evaluations of the same algorithm (OracleBRIE2 variant_b = 1 .. 5:
    float Box-Muller and / or reversed / blocked reductions) against that same oracle -- the null ensemble of the model
    variants (tests/util.py::entry_ensemble_rule)."""
    from brie_amd import _capi
    from tests import util
    mode, Kg, Kc, L, MC, target = VARIANTS[name]
    P = util.problem(Nc, Ng, Kc, L, seed=data_seed, theta=3.0)
    if Kc >= 9:
        P["Xc"] = (P["Xc"] * 0.3).astype(np.float32)       # many N(0,1) features: keep the prior mean inside the clip range
    P["Xg"] = np.random.default_rng(5 + data_seed).standard_normal((Ng, Kg)).astype(np.float32) if data_seed != 37 else \
        np.random.default_rng(5).standard_normal((Ng, Kg)).astype(np.float32)
    runs = {"o32": util.oracle_model(P, Nc, Ng, Kc, seed, np.float32, Kg=Kg, mode=mode)}
    if null:
        for v in NULL_MEMBERS:
            runs["o32b%d" % v] = util.oracle_model(P, Nc, Ng, Kc, seed, np.float32, Kg=Kg, mode=mode, variant_b=v)
    else:
        runs["o64"] = util.oracle_model(P, Nc, Ng, Kc, seed, np.float64, Kg=Kg, mode=mode)
    sh = util.device_shard(P, Nc, Ng, Kc, seed, Kg=Kg, mode=mode)
    sh.set_target(target)
    t0 = time.time()
    for n, lr in util.staged_schedule(min_iter):
        for o in runs.values():
            o.reset_optimizer()
            o.minimize(P["counts_pc"], P["Xc"], n, lr, MC, target=target)
        sh.reset_optimizer()
        sh.step(n, lr, MC, trace=False)
    if target == "marginLik":        # the posterior is not fitted: compare the prior mean's Psi, sigmoid(Xc W + Wg Xg^T + b)
        def prior_psi(W, Wg, b):
            m = np.asarray(P["Xc"], np.float64) @ np.asarray(W, np.float64) + np.asarray(b, np.float64)
            if Kg:
                m = m + np.asarray(Wg, np.float64) @ np.asarray(P["Xg"], np.float64).T
            return 1.0 / (1.0 + np.exp(-m))
        psi = {k: prior_psi(o.Wc_loc, o.Wg_loc, o.intercept) for k, o in runs.items()}
        psi["hip"] = prior_psi(sh.read(_capi.WC_LOC), sh.read(_capi.WG_LOC), sh.read(_capi.INTERCEPT))
    else:
        psi = {k: np.asarray(o.Psi, np.float64) for k, o in runs.items()}
        psi["hip"] = sh.read(_capi.PSI).astype(np.float64)
    sh.close()
    out = {"variant": name, "mode": mode, "Kg": Kg, "Kc": Kc, "count_layers": L, "MC_size": MC, "target": target,
           "model_seed": seed, "data_seed": data_seed, "shape": [Nc, Ng], "steps": 6 * int(min_iter / 6), "seconds": time.time() - t0,
           "compared": "sigmoid(prior mean)" if target == "marginLik" else "Psi"}
    pairs = [("hip_vs_o32", "hip", "o32")] + [("o32b%d_vs_o32" % v, "o32b%d" % v, "o32") for v in NULL_MEMBERS] if null else \
        [("hip_vs_o64", "hip", "o64"), ("o32_vs_o64", "o32", "o64")]
    for key, a, b in pairs:
        d = np.abs(psi[a] - psi[b])
        ex = d > 1e-4
        out[key] = {"max": float(d.max()), "p99": float(np.percentile(d, 99)), "p99.9": float(np.percentile(d, 99.9)),
                    "n_gt_1e-4": int(ex.sum()), "frac_gt_1e-4": float(ex.mean()),
                    "columns_with_more_than_5": int((ex.sum(0) > 5).sum()), "rows_with_more_than_5": int((ex.sum(1) > 5).sum()),
                    "n_outside_those_lines": int(ex[np.ix_(ex.sum(1) <= 5, ex.sum(0) <= 5)].sum())}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "psi_delta_variants_r03.json"))
    ap.add_argument("--variants", default=",".join(VARIANTS))
    ap.add_argument("--seed", type=int, default=41)
    ap.add_argument("--data-seed", type=int, default=37)
    args = ap.parse_args()
    res = {"definition": __doc__.split("\n\n")[0], "cases": {}}
    for name in [v for v in args.variants.split(",") if v]:
        r = run_variant(name, seed=args.seed, data_seed=args.data_seed)
        res["cases"][name] = r
        print("%-32s HIP-o64 max %.2e p99 %.2e n>1e-4 %6d cols %3d rows %3d rest %4d | o32-o64 max %.2e p99 %.2e n %6d cols %3d rows %3d rest %4d (%.0f s)" % (
            name, r["hip_vs_o64"]["max"], r["hip_vs_o64"]["p99"], r["hip_vs_o64"]["n_gt_1e-4"], r["hip_vs_o64"]["columns_with_more_than_5"],
            r["hip_vs_o64"]["rows_with_more_than_5"], r["hip_vs_o64"]["n_outside_those_lines"],
            r["o32_vs_o64"]["max"], r["o32_vs_o64"]["p99"], r["o32_vs_o64"]["n_gt_1e-4"], r["o32_vs_o64"]["columns_with_more_than_5"],
            r["o32_vs_o64"]["rows_with_more_than_5"], r["o32_vs_o64"]["n_outside_those_lines"], r["seconds"]), flush=True)
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
