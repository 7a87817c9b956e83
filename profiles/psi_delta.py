#!/usr/bin/env python
"""PSI-delta evidence: how far is the HIP path from the CPU reference, and how far is the reference's own
fp32 precision from the precision-independent answer, after the FULL default schedules?

    python profiles/psi_delta.py --out profiles/history/psi_delta_r02.json            (GPU box)
    python profiles/psi_delta.py --oracles-only                               (no GPU: fills the oracle cache)

For every case the same seeded problem (same init, same Philox noise stream) is run through
  hip        libbrie_amd.so as built by default (hardware transcendentals, v_rcp/v_sqrt in the Adam update)
  hip_adam   -DBRIE_STRICT_ADAM=1: IEEE division + square root in the Adam update only
  hip_strict -DBRIE_FAST_MATH=0: ocml transcendentals + IEEE division / square root everywhere
  o32        oracle/brie_oracle.c in fp32 (the reference's precision, operation by operation)
  o64        the same code in fp64 (precision-independent answer)
and |dPsi| is summarised for the pairs that matter: max, p99, p99.9, fraction > 1e-4, the same restricted to
entries with c1 + c2 > 0 ("covered"), and which share of the exceedances are zero-coverage entries.
Schedules: BRIE2.fit defaults (6 x 166 = 996 steps, MC_size 1; model_TFProb.py:214-241) and the brie-quant
defaults (6 x 833 = 4998 steps, MC_size 3; bin/quant.py:173-177).  Genes are independent (model_wrap.py:241),
so a gene sample over ALL cells of configs[1] / configs[2] is exact for those genes.
The oracle is the checker here, never the thing measured.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
CACHE = os.path.join(ROOT, "profiles", "_psi_cache")

CASES = {
    # name: Nc, Ng, Kc, L, theta, min_iter, MC
    "c1_api": dict(Nc=200, Ng=500, Kc=1, L=2, theta=3.0, min_iter=1000, MC=1,
                   desc="configs[0] 200 x 500 (+1 covariate), BRIE2.fit default schedule: 996 steps, MC_size 1"),
    "c1_kc0_api": dict(Nc=200, Ng=500, Kc=0, L=2, theta=3.0, min_iter=1000, MC=1,
                       desc="configs[0] 200 x 500, no covariate, 996 steps, MC_size 1"),
    "c1_cli": dict(Nc=200, Ng=500, Kc=1, L=2, theta=3.0, min_iter=5000, MC=3,
                   desc="configs[0] 200 x 500 (+1 covariate), brie-quant default schedule: 4998 steps, MC_size 3"),
    "c2_api": dict(Nc=10000, Ng=64, Kc=1, L=3, theta=1.5, min_iter=1000, MC=1,
                   desc="configs[1] 10k cells, effLen, Kc=1: 64-gene sample over all cells, 996 steps, MC_size 1"),
    "c2_cli": dict(Nc=10000, Ng=32, Kc=1, L=3, theta=1.5, min_iter=5000, MC=3,
                   desc="configs[1]: 32-gene sample over all cells, 4998 steps, MC_size 3"),
    "c3_api": dict(Nc=50000, Ng=32, Kc=3, L=2, theta=1.5, min_iter=1000, MC=1,
                   desc="configs[2] 50k cells, Kc=3: 32-gene sample over all cells, 996 steps, MC_size 1"),
    "c3_cli": dict(Nc=50000, Ng=16, Kc=3, L=2, theta=1.5, min_iter=5000, MC=3,
                   desc="configs[2]: 16-gene sample over all cells, 4998 steps, MC_size 3"),
    # round 3: the samples the parity rule is frozen on -- hundreds of genes over ALL cells
    "c3_api_512": dict(Nc=50000, Ng=512, Kc=3, L=2, theta=1.5, min_iter=1000, MC=1,
                       desc="configs[2] 50k cells, Kc=3: 512 genes of the recipe over all cells, 996 steps, MC_size 1"),
    "c2_api_512": dict(Nc=10000, Ng=512, Kc=1, L=3, theta=1.5, min_iter=1000, MC=1,
                       desc="configs[1] 10k cells, effLen, Kc=1: 512 genes over all cells, 996 steps, MC_size 1"),
    "c3_cli_128": dict(Nc=50000, Ng=128, Kc=3, L=2, theta=1.5, min_iter=5000, MC=3,
                       desc="configs[2]: 128 genes over all cells, brie-quant schedule: 4998 steps, MC_size 3"),
    "c2_cli_128": dict(Nc=10000, Ng=128, Kc=1, L=3, theta=1.5, min_iter=5000, MC=3,
                       desc="configs[1]: 128 genes over all cells, brie-quant schedule: 4998 steps, MC_size 3"),
    # round 3, AFTER the rule was frozen on the seven cases above: other data, other initial state, other noise stream
    # (out-of-sample check of tests/util.py::psi_parity_rule -- nothing was tuned on these)
    "c3_api_512_s2": dict(Nc=50000, Ng=512, Kc=3, L=2, theta=1.5, min_iter=1000, MC=1, data_seed=8675309, seed=23,
                          desc="configs[2] shape, OTHER data seed / model seed: 512 genes over all cells, 996 steps, MC_size 1"),
    "c2_api_512_s2": dict(Nc=10000, Ng=512, Kc=1, L=3, theta=1.5, min_iter=1000, MC=1, data_seed=8675309, seed=23,
                          desc="configs[1] shape, OTHER data seed / model seed: 512 genes over all cells, 996 steps, MC_size 1"),
    "c2_cli_128_s2": dict(Nc=10000, Ng=128, Kc=1, L=3, theta=1.5, min_iter=5000, MC=3, data_seed=8675309, seed=23,
                          desc="configs[1] shape, OTHER seeds: 128 genes over all cells, 4998 steps, MC_size 3"),
    "c3_cli_128_s2": dict(Nc=50000, Ng=128, Kc=3, L=2, theta=1.5, min_iter=5000, MC=3, data_seed=8675309, seed=23,
                          desc="configs[2] shape, OTHER seeds: 128 genes over all cells, 4998 steps, MC_size 3"),
    # ... and a shape none of the configs has (20k cells, 2 covariates, effLen + ambiguous layer), third set of seeds
    "mid_api_256_s3": dict(Nc=20000, Ng=256, Kc=2, L=3, theta=2.0, min_iter=1000, MC=1, data_seed=424243, seed=37,
                           desc="20k cells x 256 genes, effLen, Kc=2 (no config's shape), third data / model seed, 996 steps, MC_size 1"),
    "mid_cli_96_s3": dict(Nc=20000, Ng=96, Kc=2, L=2, theta=2.0, min_iter=5000, MC=3, data_seed=424243, seed=37,
                          desc="20k cells x 96 genes, 2 layers, Kc=2, third seeds, 4998 steps, MC_size 3"),
    # round 3, second held-out set: generated AFTER the rule's revision 2 (gene-level clusters counted as gene-level
    # events) -- fourth set of seeds, nothing tuned on these either
    "c1_cli_s4": dict(Nc=200, Ng=500, Kc=1, L=2, theta=3.0, min_iter=5000, MC=3, data_seed=99991, seed=41,
                      desc="configs[0] shape, fourth seeds, brie-quant schedule: 4998 steps, MC_size 3"),
    "c2_api_512_s4": dict(Nc=10000, Ng=512, Kc=1, L=3, theta=1.5, min_iter=1000, MC=1, data_seed=99991, seed=41,
                          desc="configs[1] shape, fourth seeds: 512 genes over all cells, 996 steps, MC_size 1"),
    "c2_cli_128_s4": dict(Nc=10000, Ng=128, Kc=1, L=3, theta=1.5, min_iter=5000, MC=3, data_seed=99991, seed=41,
                          desc="configs[1] shape, fourth seeds: 128 genes over all cells, 4998 steps, MC_size 3"),
    "mid_api_256_s4": dict(Nc=20000, Ng=256, Kc=2, L=3, theta=2.0, min_iter=1000, MC=1, data_seed=99991, seed=41,
                           desc="20k cells x 256 genes, effLen, Kc=2, fourth seeds, 996 steps, MC_size 1"),
    "mid_cli_96_s4": dict(Nc=20000, Ng=96, Kc=2, L=2, theta=2.0, min_iter=5000, MC=3, data_seed=99991, seed=41,
                          desc="20k cells x 96 genes, 2 layers, Kc=2, fourth seeds, 4998 steps, MC_size 3"),
    "c3_api_256_s4": dict(Nc=50000, Ng=256, Kc=3, L=2, theta=1.5, min_iter=1000, MC=1, data_seed=99991, seed=41,
                          desc="configs[2] shape, fourth seeds: 256 genes over all cells, 996 steps, MC_size 1"),
    "c3_cli_64_s4": dict(Nc=50000, Ng=64, Kc=3, L=2, theta=1.5, min_iter=5000, MC=3, data_seed=99991, seed=41,
                         desc="configs[2] shape, fourth seeds: 64 genes over all cells, 4998 steps, MC_size 3"),
    # round 5: ONE new held-out set of seeds per shape for the pre-registered null ensemble (profiles/psi_ensemble.py;
    # tests/golden/psi_ensemble_manifest.json) -- chosen before any run of either side
    "c2_cli_64_s5": dict(Nc=10000, Ng=64, Kc=1, L=3, theta=1.5, min_iter=5000, MC=3, data_seed=5550123, seed=59,
                         desc="configs[1] shape, fifth seeds: 64 genes over all cells, 4998 steps, MC_size 3"),
    "c3_cli_64_s5": dict(Nc=50000, Ng=64, Kc=3, L=2, theta=1.5, min_iter=5000, MC=3, data_seed=5550123, seed=59,
                         desc="configs[2] shape, fifth seeds: 64 genes over all cells, 4998 steps, MC_size 3"),
}
HELD_OUT_2 = ("c1_cli_s4", "c2_api_512_s4", "c2_cli_128_s4", "mid_api_256_s4", "mid_cli_96_s4", "c3_api_256_s4", "c3_cli_64_s4")
HELD_OUT = ("c2_api_512_s2", "c3_api_512_s2", "c2_cli_128_s2", "c3_cli_128_s2", "mid_api_256_s3", "mid_cli_96_s3")
R03 = ("c1_api", "c1_kc0_api", "c1_cli", "c2_api_512", "c3_api_512", "c2_cli_128", "c3_cli_128")
PARAMS = ("Wc_loc", "intercept", "sigma_log")
QUICK = ("c1_api", "c1_kc0_api", "c2_api", "c3_api")
SEED = 11
VARIANTS = {"hip": [], "hip_adam": ["BRIE_STRICT_ADAM=1"], "hip_strict": ["BRIE_FAST_MATH=0"]}


def problem(case):
    from tests import util
    c = CASES[case]
    kw = {"seed": c["data_seed"]} if "data_seed" in c else {}
    return util.problem(c["Nc"], c["Ng"], c["Kc"], c["L"], theta=c["theta"], **kw), c


def model_seed(case):
    return CASES[case].get("seed", SEED)


def schedule(min_iter):
    from oracle.brie_oracle import LEARNING_RATES
    return [(int(min_iter / 6), lr) for lr in LEARNING_RATES]


def run_oracle(case, dtype, want_params=False):
    """Psi of the C restatement in `dtype` after the case's full staged schedule (cached)."""
    os.makedirs(CACHE, exist_ok=True)
    path = os.path.join(CACHE, "%s_%s.npz" % (case, np.dtype(dtype).name))
    if os.path.exists(path):
        z = np.load(path)
        if all(k in z.files for k in PARAMS) or not want_params:
            return {k: z[k] for k in z.files if k in PARAMS + ("psi",)}
    from oracle.c_oracle import COracle
    P, c = problem(case)
    t0 = time.time()
    o = COracle(P["counts_pc"], P["Xc"], effLen=P["effLen"], seed=model_seed(case), dtype=dtype)
    for n, lr in schedule(c["min_iter"]):
        o.reset_optimizer()
        o.minimize(n, lr, c["MC"])
    # Psi kept as float32: rounding the fp64 answer to fp32 moves it by <= 6e-8, three orders below the 1e-4 that is
    # counted, and the cache (which travels to the GPU box) stays at 4 bytes per entry
    out = {"psi": np.asarray(o.Psi, np.float32), "Wc_loc": np.asarray(o.Wc_loc, np.float64),
           "intercept": np.asarray(o.intercept, np.float64), "sigma_log": np.asarray(o.sigma_log, np.float64)}
    np.savez(path, seconds=time.time() - t0, **out)
    print("oracle %s %s: %.1f s" % (case, np.dtype(dtype).name, time.time() - t0), flush=True)
    return out


def run_hip_worker(case, out):
    """(subprocess, BRIE_AMD_LIB selects the build) Psi of the HIP path after the full staged schedule."""
    from brie_amd import _capi
    from tests import util
    P, c = problem(case)
    sh = util.device_shard(P, c["Nc"], c["Ng"], c["Kc"], model_seed(case))
    t0 = time.time()
    for n, lr in schedule(c["min_iter"]):
        sh.reset_optimizer()
        sh.step(n, lr, c["MC"], trace=False)
    psi = sh.read(_capi.PSI)
    np.savez(out, psi=psi, seconds=time.time() - t0, Wc_loc=sh.read(_capi.WC_LOC),
             intercept=sh.read(_capi.INTERCEPT).reshape(-1), sigma_log=sh.read(_capi.SIGMA_LOG).reshape(-1))
    sh.close()


def summary(a, b, covered):
    d = np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))
    ex = d > 1e-4
    dc = d[covered]
    out = {"max": float(d.max()), "p99": float(np.percentile(d, 99)), "p99.9": float(np.percentile(d, 99.9)),
           "frac_gt_1e-4": float(ex.mean()), "n_gt_1e-4": int(ex.sum()), "n": int(d.size),
           "covered": {"max": float(dc.max()), "p99": float(np.percentile(dc, 99)),
                       "p99.9": float(np.percentile(dc, 99.9)), "frac_gt_1e-4": float((dc > 1e-4).mean()),
                       "n": int(dc.size)},
           "share_of_exceedances_with_zero_coverage": float((ex & ~covered).sum() / max(1, ex.sum()))}
    return out


def build_variants(names=None):
    from brie_amd.build import compile_library, LIB_DIR
    paths = {}
    for name, defs in VARIANTS.items():
        if names is not None and name not in names:
            continue
        if not defs:
            paths[name] = compile_library()
            continue
        out = os.path.join(LIB_DIR, "libbrie_amd_%s.so" % name)
        src_t = max(os.path.getmtime(os.path.join(ROOT, "brie_amd", "csrc", f))
                    for f in os.listdir(os.path.join(ROOT, "brie_amd", "csrc")))
        if not os.path.exists(out) or os.path.getmtime(out) < src_t:
            compile_library(out=out, defines=defs)
        paths[name] = out
    return paths


def per_gene(psi, par, covered):
    """Where the exceedances sit: per gene, how many cells lie beyond 1e-4 for HIP-vs-o64, o32-vs-o64 and HIP-vs-o32, and
    whether one of the gene's OWN parameters (Wc_loc column, intercept, sigma_log) moved by more than 1e-3 between two
    of the three runs -- a shifted per-gene parameter moves all of the gene's cells at once (the per-gene clusters)."""
    def ex(a, b):
        return (np.abs(np.asarray(psi[a], np.float64) - np.asarray(psi[b], np.float64)) > 1e-4)

    def shift(a, b):
        w = np.abs(np.asarray(par[a]["Wc_loc"], np.float64) - np.asarray(par[b]["Wc_loc"], np.float64))
        w = w.max(0) if w.size else np.zeros(psi[a].shape[1])
        return np.maximum(w, np.maximum(
            np.abs(np.asarray(par[a]["intercept"], np.float64).ravel() - np.asarray(par[b]["intercept"], np.float64).ravel()),
            np.abs(np.asarray(par[a]["sigma_log"], np.float64).ravel() - np.asarray(par[b]["sigma_log"], np.float64).ravel())))
    e_h, e_o, e_ho = ex("hip", "o64"), ex("o32", "o64"), ex("hip", "o32")
    s_h, s_o, s_ho = shift("hip", "o64"), shift("o32", "o64"), shift("hip", "o32")
    n_h, n_o = int(e_h.sum()), int(e_o.sum())
    moved_h, moved_o = s_h > 1e-3, s_o > 1e-3
    out = {
        "n_entries": int(e_h.size), "n_genes": int(e_h.shape[1]),
        "exceed_hip_vs_o64": n_h, "exceed_o32_vs_o64": n_o, "exceed_hip_vs_o32": int(e_ho.sum()),
        "ratio_hip_over_o32": n_h / max(1, n_o),
        "exceed_covered_hip": int((e_h & covered).sum()), "exceed_covered_o32": int((e_o & covered).sum()),
        "ratio_covered_hip_over_o32": int((e_h & covered).sum()) / max(1, int((e_o & covered).sum())),
        "genes_with_an_exceedance": {"hip": int((e_h.sum(0) > 0).sum()), "o32": int((e_o.sum(0) > 0).sum())},
        "genes_whose_own_parameter_moved_gt_1e-3": {"hip_vs_o64": int(moved_h.sum()), "o32_vs_o64": int(moved_o.sum()),
                                                    "hip_vs_o32": int((s_ho > 1e-3).sum())},
        "share_of_exceedances_in_those_genes": {"hip": float(e_h[:, moved_h].sum() / max(1, n_h)),
                                                "o32": float(e_o[:, moved_o].sum() / max(1, n_o))},
        "max_per_gene_parameter_shift": {"hip_vs_o64": float(s_h.max()), "o32_vs_o64": float(s_o.max()),
                                         "hip_vs_o32": float(s_ho.max())},
        "per_gene": {"exceed_hip": e_h.sum(0).astype(int).tolist(), "exceed_o32": e_o.sum(0).astype(int).tolist(),
                     "exceed_hip_vs_o32": e_ho.sum(0).astype(int).tolist(),
                     "param_shift_hip_vs_o64": [float("%.3g" % x) for x in s_h],
                     "param_shift_o32_vs_o64": [float("%.3g" % x) for x in s_o]},
    }
    return out


def util_params(p):
    return {"Wc_loc": np.asarray(p["Wc_loc"], np.float64), "intercept": np.asarray(p["intercept"], np.float64).reshape(-1),
            "sigma_log": np.asarray(p["sigma_log"], np.float64).reshape(-1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "psi_delta.json"),
                    help="(the committed profiles/history/psi_delta_r03.json is the evidence revision 1 of the rule was frozen on: not a default target)")
    ap.add_argument("--cases", default=",".join(R03))
    ap.add_argument("--variants", default="hip")
    ap.add_argument("--oracles-only", action="store_true")
    ap.add_argument("--build-only", action="store_true")
    ap.add_argument("--slice-cache", default=None, metavar="CASE:N",
                    help="write the first N genes of a case's oracle cache as <case>_firstN_<dtype>.npz (genes are "
                         "independent; for tests whose full cache does not fit the GPU boxes' 512-MiB snapshot)")
    ap.add_argument("--worker", default=None)
    ap.add_argument("--worker-out", default=None)
    args = ap.parse_args()
    if args.worker:
        run_hip_worker(args.worker, args.worker_out)
        return
    if args.slice_cache:
        case, n = args.slice_cache.split(":")
        n = int(n)
        for dt in ("float32", "float64"):
            z = np.load(os.path.join(CACHE, "%s_%s.npz" % (case, dt)))
            np.savez(os.path.join(CACHE, "%s_first%d_%s.npz" % (case, n, dt)), psi=z["psi"][:, :n], Wc_loc=z["Wc_loc"][:, :n],
                     intercept=np.asarray(z["intercept"]).reshape(-1)[:n], sigma_log=np.asarray(z["sigma_log"]).reshape(-1)[:n],
                     seconds=z["seconds"])
        return
    cases = [c for c in args.cases.split(",") if c]
    if args.oracles_only:
        for case in cases:
            run_oracle(case, np.float32, want_params=True)
            run_oracle(case, np.float64, want_params=True)
        return
    libs = build_variants([v for v in args.variants.split(",") if v])
    if args.build_only:
        print(libs)
        return
    result = {"seed": SEED, "definition": __doc__.split("\n\n")[2].strip(), "cases": {}}
    for case in cases:
        P, c = problem(case)
        covered = (np.asarray(P["counts"][0]) + np.asarray(P["counts"][1])) > 0
        par = {"o32": run_oracle(case, np.float32, want_params=True), "o64": run_oracle(case, np.float64, want_params=True)}
        psi = {k: v["psi"] for k, v in par.items()}
        secs = {}
        for v in [v for v in args.variants.split(",") if v]:
            tmp = os.path.join(CACHE, "_%s_%s.npz" % (case, v))
            env = dict(os.environ, BRIE_AMD_LIB=libs[v])
            subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", case, "--worker-out", tmp],
                           check=True, env=env)
            z = np.load(tmp)
            psi[v], secs[v] = z["psi"], float(z["seconds"])
            par[v] = {k: z[k] for k in PARAMS}
            os.remove(tmp)
        pairs = [(v, "o64") for v in psi if v.startswith("hip")] + [(v, "o32") for v in psi if v.startswith("hip")] + \
                [("o32", "o64")] + [("hip", v) for v in psi if v.startswith("hip_")]
        entry = {"desc": c["desc"], "model_seed": model_seed(case), "data_seed": c.get("data_seed", 20240617), "shape": [c["Nc"], c["Ng"]], "Kc": c["Kc"], "count_layers": c["L"],
                 "steps": 6 * int(c["min_iter"] / 6), "MC_size": c["MC"],
                 "zero_coverage_fraction": float(1 - covered.mean()), "hip_seconds": secs, "pairs": {}}
        for a, b in pairs:
            entry["pairs"]["%s_vs_%s" % (a, b)] = summary(psi[a], psi[b], covered)
        if "hip" in psi:
            entry["where_the_exceedances_sit"] = per_gene(psi, par, covered)
            from tests import util
            try:                                     # the frozen rule of the test-suite on this case
                entry["parity_rule"] = dict(util.psi_parity_rule(psi, {k: util_params(par[k]) for k in ("hip", "o32", "o64")}, case),
                                            holds=True)
            except AssertionError as exc:
                entry["parity_rule"] = {"holds": False, "violated": repr(exc)}
        result["cases"][case] = entry
        f = entry["pairs"]
        g = entry.get("where_the_exceedances_sit", {})
        print("%-11s hip-o64 max %.2e p99.9 %.2e frac>1e-4 %.2e | o32-o64 max %.2e p99.9 %.2e frac %.2e | "
              "HIP/o32 exceedances %.2f (covered %.2f)" % (
                  case, f["hip_vs_o64"]["max"], f["hip_vs_o64"]["p99.9"], f["hip_vs_o64"]["frac_gt_1e-4"],
                  f["o32_vs_o64"]["max"], f["o32_vs_o64"]["p99.9"], f["o32_vs_o64"]["frac_gt_1e-4"],
                  g.get("ratio_hip_over_o32", float("nan")), g.get("ratio_covered_hip_over_o32", float("nan"))), flush=True)
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as fh:            # after every case: a cut-off call still leaves the finished ones
            json.dump(result, fh, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
