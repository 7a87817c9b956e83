"""Shared helpers for the parity tests: run the same problem through the CPU
oracle (oracle/) and through the HIP path (C ABI via brie_amd._capi)."""
import numpy as np

from oracle.brie_oracle import OracleBRIE2, add_pseudo_count, LEARNING_RATES
from oracle.synth import make_problem

STATE_KEYS = ("Z_loc", "Z_std_log", "Wc_loc", "Wg_loc", "intercept", "sigma_log")


def problem(Nc, Ng, Kc, L, seed=20240617, theta=1.5):
    P = make_problem(Nc, Ng, Kc=Kc, L=L, seed=seed, theta=theta)
    P["counts_pc"] = add_pseudo_count(P["counts"], 0.01)
    return P


def oracle_model(P, Nc, Ng, Kc, seed, dtype=np.float32, gene_offset=0, intercept=None, sigma=None, Kg=0,
                 mode='gene'):
    o = OracleBRIE2(Nc, Ng, Kc, effLen=P["effLen"], seed=seed, dtype=dtype,
                    gene_offset=gene_offset, intercept=intercept, sigma=sigma, Kg=Kg, intercept_mode=mode)
    o.Xg = P.get("Xg")
    return o


def device_shard(P, Nc, Ng, Kc, seed, gene_offset=0, intercept=None, sigma=None, pseudo=0.01, storage=None, Kg=0,
                 mode='gene'):
    from brie_amd import _capi
    L = len(P["counts"])
    sh = _capi.Shard(Nc, Ng, Kc, n_layers=L, has_efflen=P["effLen"] is not None,
                     train_intercept=intercept is None, train_sigma=sigma is None,
                     seed=seed, gene_offset=gene_offset, Kg=Kg, intercept_mode=1 if mode == 'cell' else 0)
    if storage == "f32":
        sh.set_count_storage(1)
    for l in range(L):
        sh.upload(_capi.COUNT1 + l, P["counts"][l])
    if pseudo:
        sh.add_pseudo_count(pseudo)
    if P["effLen"] is not None:
        sh.upload(_capi.EFFLEN, P["effLen"])
    if Kc > 0:
        sh.upload(_capi.XC, P["Xc"])
    if Kg > 0:
        sh.upload(_capi.XG, P["Xg"])
    sh.init_state(intercept, sigma)
    return sh


def device_state(sh):
    from brie_amd import _capi
    return {"Z_loc": sh.read(_capi.Z_LOC), "Z_std_log": sh.read(_capi.Z_STD_LOG),
            "Wc_loc": sh.read(_capi.WC_LOC), "Wg_loc": sh.read(_capi.WG_LOC),
            "intercept": sh.read(_capi.INTERCEPT),
            "sigma_log": sh.read(_capi.SIGMA_LOG)}


def oracle_state(o):
    return {k: np.asarray(getattr(o, k)) for k in STATE_KEYS}


def max_abs_diff(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b))) if a.size else 0.0


def staged_schedule(min_iter):
    return [(int(min_iter / 6), lr) for lr in LEARNING_RATES]
