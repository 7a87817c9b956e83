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


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def part_c():
    """Reference host helpers that import without TF/pysam by file path:
    brie/utils/preprocessing.py:filter_genes (5-83) and brie/utils/base_utils.py:match (5-59)."""
    pp = _load("/root/reference/brie/utils/preprocessing.py", "ref_pp")
    bu = _load("/root/reference/brie/utils/base_utils.py", "ref_bu")
    rng = np.random.default_rng(7)

    class Stub(object):          # the slice of AnnData filter_genes touches
        def __init__(self, layers):
            self.layers = layers
            self.shape = layers['isoform1'].shape
            self.n_vars = self.shape[1]
            self.var = {}
            self.kept = None

        def copy(self):
            return Stub({k: v.copy() for k, v in self.layers.items()})

        def _inplace_subset_var(self, mask):
            self.kept = mask.copy()
            self.layers = {k: v[:, mask] for k, v in self.layers.items()}

    Nc, Ng = 60, 80
    depth = rng.lognormal(0, 1.2, Ng)
    psi = rng.beta(0.4, 0.4, Ng)
    N = rng.poisson(depth[None, :] * np.ones((Nc, 1)))
    c1 = rng.binomial(N, psi[None, :])
    layers = {'isoform1': c1.astype(float), 'isoform2': (N - c1).astype(float),
              'ambiguous': rng.poisson(0.5 * depth[None, :] * np.ones((Nc, 1))).astype(float)}
    cases = []
    for kw in (dict(min_counts=50, min_counts_uniq=10, min_cells_uniq=30, min_MIF_uniq=0.001),
               dict(min_counts=0, min_cells=5, min_counts_uniq=0, min_cells_uniq=0, min_MIF_uniq=0.05),
               dict(min_counts=200, min_counts_uniq=100, min_cells_uniq=50, min_MIF_uniq=0.2)):
        st = Stub({k: v.copy() for k, v in layers.items()})
        out = pp.filter_genes(st, copy=True, **kw)
        cases.append(dict(kw=kw, kept=out.kept, n_counts=out.var['n_counts'], n_counts_uniq=out.var['n_counts_uniq']))
    ref_ids = np.array(["c%03d" % i for i in rng.permutation(50)])
    new_ids = np.array(["c%03d" % i for i in rng.permutation(70)[:40]])
    m = bu.match(ref_ids, new_ids)
    # duplicated ref ids (uniq_ref_only=True, the default): 12 entries -- numpy's argsort is an insertion sort
    # below 16 elements, i.e. stable, so the reference's answer is well defined here
    dup_ref = np.array(["b", "a", "c", "a", "d", "b", "e", "a", "f", "g", "c", "h"])
    dup_new = np.array(["h", "c", "a", "x", "b", "e"])
    md = bu.match(dup_ref, dup_new)
    md_all = bu.match(dup_ref, dup_new, uniq_ref_only=False)
    # copy=False: in place, returns None
    st = Stub({k: v.copy() for k, v in layers.items()})
    ret = pp.filter_genes(st, copy=False, min_counts=50, min_counts_uniq=10, min_cells_uniq=30)
    assert ret is None
    inplace = dict(kept=st.kept, n_counts=st.var['n_counts'], n_left=st.layers['isoform1'].shape[1])
    np.savez_compressed(os.path.join(HERE, "ref_match_dup_inplace.npz"), dup_ref=dup_ref, dup_new=dup_new,
                        dup_idx=np.array([-1 if x is None else x for x in md], dtype=int),
                        dup_idx_all=np.array([-1 if x is None else x for x in md_all], dtype=int),
                        inplace=np.array([inplace], dtype=object))
    np.savez_compressed(os.path.join(HERE, "ref_filter_match.npz"), isoform1=layers['isoform1'],
                        isoform2=layers['isoform2'], ambiguous=layers['ambiguous'],
                        cases=np.array(cases, dtype=object), ref_ids=ref_ids, new_ids=new_ids,
                        match_idx=np.array([-1 if x is None else x for x in m], dtype=int))


if __name__ == "__main__":
    part_a()
    part_b()
    part_c()
    print("golden fixtures written to", HERE)
