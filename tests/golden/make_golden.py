"""Generate the committed golden fixtures.  Run ONLY in the build container:

    python tests/golden/make_golden.py

Part A imports the two reference fragments that load by file path without
TensorFlow -- /root/reference/brie/models/base_model.py: `get_CI95` (29-36) and
`BRIE_base_lik` (20-27) -- evaluates them on seeded random grids and stores
inputs + outputs (data only; no reference source travels).
Part B stores fp64 oracle trajectories on tiny problems as regression pins for
oracle/brie_oracle.py itself and as targets for the HIP path.
"""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def load_reference_fragment():
    path = "/root/reference/brie/models/base_model.py"
    spec = importlib.util.spec_from_file_location("ref_base_model", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def part_a():
    ref = load_reference_fragment()
    rng = np.random.default_rng(20240617)
    # get_CI95(Psi, Z_std) -> (low, high)
    Psi = rng.uniform(0.001, 0.999, size=(64, 48))
    Z_std = np.exp(rng.normal(0, 1, size=(64, 48)))
    low, high = ref.get_CI95(Psi, Z_std)
    np.savez_compressed(os.path.join(HERE, "ref_get_CI95.npz"), Psi=Psi, Z_std=Z_std, low=low, high=high)
    # BRIE_base_lik(psi, counts, lengths) -> multinomial pmf
    n = 400
    psi = rng.uniform(0.01, 0.99, size=n)
    counts = rng.integers(0, 12, size=(n, 3))
    lengths = rng.integers(30, 400, size=(n, 3)).astype(np.float64)
    pmf = np.array([ref.BRIE_base_lik(psi[i], counts[i], lengths[i]) for i in range(n)])
    np.savez_compressed(os.path.join(HERE, "ref_BRIE_base_lik.npz"), psi=psi, counts=counts,
                        lengths=lengths, pmf=pmf)


def part_b():
    from oracle.brie_oracle import OracleBRIE2, add_pseudo_count, LEARNING_RATES
    from oracle.synth import make_problem
    cases = {"lik2_kc2": dict(Nc=48, Ng=36, Kc=2, L=2, MC=1),
             "eff3_kc1_mc3": dict(Nc=40, Ng=28, Kc=1, L=3, MC=3)}
    for name, c in cases.items():
        P = make_problem(c["Nc"], c["Ng"], Kc=c["Kc"], L=c["L"], seed=77, theta=2.0)
        cnt = add_pseudo_count(P["counts"])
        o = OracleBRIE2(c["Nc"], c["Ng"], c["Kc"], effLen=P["effLen"], seed=1234, dtype=np.float64)
        init = {k: getattr(o, k).copy() for k in ("Z_loc", "Z_std_log", "Wc_loc", "intercept", "sigma_log")}
        traces = []
        for lr in LEARNING_RATES:
            o.reset_optimizer()
            traces.append(o.minimize(cnt, P["Xc"], 20, lr, c["MC"]))
        lg = o.eval_loss_gene(cnt, P["Xc"], 10)
        out = dict(seed=1234, steps_per_stage=20, MC=c["MC"], Xc=P["Xc"], losses=np.concatenate(traces),
                   loss_gene=lg, Psi=o.Psi, Psi95CI=o.Psi95CI, sigma=o.sigma,
                   **{"init_" + k: v for k, v in init.items()},
                   **{"final_" + k: getattr(o, k) for k in init},
                   **{"count%d" % (i + 1): a for i, a in enumerate(P["counts"])})
        if P["effLen"] is not None:
            out["effLen"] = P["effLen"]
        np.savez_compressed(os.path.join(HERE, "oracle_traj_%s.npz" % name), **out)


if __name__ == "__main__":
    part_a()
    part_b()
    print("golden fixtures written to", HERE)
