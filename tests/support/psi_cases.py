"""The cases of the fit-level parity evidence: gene samples of the BASELINE configs over ALL cells under both default
schedules -- BRIE2.fit's (6 x 166 = 996 steps, MC_size 1; model_TFProb.py:214-241) and brie-quant's (6 x 833 = 4998 steps,
MC_size 3; bin/quant.py:173-177).  Genes are independent (model_wrap.py:241), so a gene sample over ALL cells of
configs[1] / configs[2] is exact for those genes.  TEST INFRASTRUCTURE: the GPU suite (tests/test_gpu_fullsize.py), the
negative controls (tests/test_rule_power.py) and the scripts under profiles/ that fill the oracle caches import from here.
"""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CACHE = os.path.join(ROOT, "profiles", "_psi_cache")        # git-ignored oracle runs (tests/golden/psi_null_caches.json)

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
    # round 5, addendum: three MORE held-out cases for the ensemble rule, registered (manifest: registered_addendum_1) after the
    # first four had been judged and before anything ran on these -- sixth seeds; the third is a shape no config has
    "c2_cli_64_s6": dict(Nc=10000, Ng=64, Kc=1, L=3, theta=1.5, min_iter=5000, MC=3, data_seed=777001, seed=67,
                         desc="configs[1] shape, sixth seeds: 64 genes over all cells, 4998 steps, MC_size 3"),
    "c3_cli_64_s6": dict(Nc=50000, Ng=64, Kc=3, L=2, theta=1.5, min_iter=5000, MC=3, data_seed=777001, seed=67,
                         desc="configs[2] shape, sixth seeds: 64 genes over all cells, 4998 steps, MC_size 3"),
    "mid_cli_64_s6": dict(Nc=20000, Ng=64, Kc=2, L=3, theta=2.0, min_iter=5000, MC=3, data_seed=777001, seed=67,
                          desc="20k cells x 64 genes, effLen, Kc=2 (no config's shape), sixth seeds, 4998 steps, MC_size 3"),
    # round 5, addendum 2: the API schedule (996 steps, MC_size 1) under the ensemble rule as well -- two held-out cases
    "c2_api_64_s7": dict(Nc=10000, Ng=64, Kc=1, L=3, theta=1.5, min_iter=1000, MC=1, data_seed=888002, seed=71,
                         desc="configs[1] shape, seventh seeds: 64 genes over all cells, 996 steps, MC_size 1"),
    "c3_api_64_s7": dict(Nc=50000, Ng=64, Kc=3, L=2, theta=1.5, min_iter=1000, MC=1, data_seed=888002, seed=71,
                         desc="configs[2] shape, seventh seeds: 64 genes over all cells, 996 steps, MC_size 1"),
    # round 5, addendum 3: the two BASELINE shapes the ensemble had not seen -- configs[4] (DMG mode: 100k cells, 2 layers, 5
    # covariates) and configs[0] (200 x 500, no covariate) -- at the brie-quant default schedule, eighth seeds, held out
    "c5_cli_64_s8": dict(Nc=100000, Ng=64, Kc=5, L=2, theta=1.5, min_iter=5000, MC=3, data_seed=999003, seed=73,
                         desc="configs[4] shape (100k cells, Kc=5, 2 layers), eighth seeds: 64 genes over all cells, 4998 steps, MC_size 3"),
    "c1_kc0_cli_s8": dict(Nc=200, Ng=500, Kc=0, L=2, theta=3.0, min_iter=5000, MC=3, data_seed=999003, seed=73,
                          desc="configs[0] shape (200 x 500, no covariate), eighth seeds, 4998 steps, MC_size 3"),
}
HELD_OUT_2 = ("c1_cli_s4", "c2_api_512_s4", "c2_cli_128_s4", "mid_api_256_s4", "mid_cli_96_s4", "c3_api_256_s4", "c3_cli_64_s4")
HELD_OUT = ("c2_api_512_s2", "c3_api_512_s2", "c2_cli_128_s2", "c3_cli_128_s2", "mid_api_256_s3", "mid_cli_96_s3")
R03 = ("c1_api", "c1_kc0_api", "c1_cli", "c2_api_512", "c3_api_512", "c2_cli_128", "c3_cli_128")
PARAMS = ("Wc_loc", "intercept", "sigma_log")
QUICK = ("c1_api", "c1_kc0_api", "c2_api", "c3_api")
SEED = 11


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


def util_params(p):
    return {"Wc_loc": np.asarray(p["Wc_loc"], np.float64), "intercept": np.asarray(p["intercept"], np.float64).reshape(-1),
            "sigma_log": np.asarray(p["sigma_log"], np.float64).reshape(-1)}
