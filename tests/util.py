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


def psi_parity_assert(d, d32, what=""):
    """THE parity rule of the floating-point path (north star: PSI within 1e-4).

    d   = |Psi_hip - Psi_o64|   HIP path vs the fp64 oracle (the precision-independent answer)
    d32 = |Psi_o32 - Psi_o64|   the reference's own precision (fp32 oracle) vs the same answer, same entries

    Keras Adam moves every entry by lr * m / (sqrt(v) + eps): where a gradient passes through ~0 the SIGN of an
    fp32 rounding error decides an O(lr) move (a fresh optimiser's first step is +-lr whatever |g| is), so any two
    fp32 evaluations of the reference's arithmetic -- TF on another CPU included -- differ by more than 1e-4 on a
    fraction of the entries.  Measured after the full default schedules (profiles/psi_delta_r02.json): that fraction
    is the same for HIP-vs-fp64 and fp32-oracle-vs-fp64 (C3: 3e-5 vs 1e-5, >90 % of them zero-coverage entries;
    the 200-cell configs[0]: 6e-3 vs 5e-3 at 996 steps, 2.4e-2 vs 2.4e-2 at 4998), and strict IEEE math on the
    device does not change it.  So parity = "as close to the precision-independent answer as the reference's own
    fp32 arithmetic gets on the same trajectory":
      1. bulk:        p99(d) <= max(1e-4, 1.5 p99(d32));
      2. exceedances: #(d > 1e-4) <= 3 #(d32 > 1e-4) + max(5e-5 n, 20)   (the exceedances come in per-gene clusters --
                      a gene's Wc_loc / intercept / sigma trajectory shifts all its cells -- so between two fp32
                      evaluations their count fluctuates far more than Poisson: factor 3 and a small floor);
      3. worst entry: max d <= max(2e-3, 3 max d32)   (a fifth of what ONE flipped +-lr step can do: 0.25 * 2 * 0.02)."""
    d, d32 = np.asarray(d, np.float64).ravel(), np.asarray(d32, np.float64).ravel()
    n, n32 = int((d > 1e-4).sum()), int((d32 > 1e-4).sum())
    assert np.percentile(d, 99) <= max(1e-4, 1.5 * np.percentile(d32, 99)), \
        (what, "p99", float(np.percentile(d, 99)), float(np.percentile(d32, 99)))
    assert n <= 3 * n32 + max(5e-5 * d.size, 20), (what, "entries beyond 1e-4: HIP %d, fp32 oracle %d of %d" % (n, n32, d.size))
    assert d.max() <= max(2e-3, 3 * d32.max()), (what, "max", float(d.max()), float(d32.max()))
    return {"max": float(d.max()), "p99": float(np.percentile(d, 99)), "p99.9": float(np.percentile(d, 99.9)),
            "frac_gt_1e-4": n / d.size, "fp32_oracle": {"max": float(d32.max()), "p99": float(np.percentile(d32, 99)),
                                                         "p99.9": float(np.percentile(d32, 99.9)),
                                                         "frac_gt_1e-4": n32 / d32.size}}
