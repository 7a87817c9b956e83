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


PSI_TOL = 1e-4                  # north star: PSI within 1e-4 of the CPU path
GENE_SHIFT = 4 * PSI_TOL        # a shift of a gene's OWN parameter (Wc_loc column, intercept, sigma_log) that moves Psi by
                                # PSI_TOL where sigmoid' is largest (1/4): such a gene is "displaced" as a whole


def run_params(obj, cols=None):
    """The per-gene parameters of a run -- an oracle (OracleBRIE2 / COracle) or a device shard -- as float64:
    dict(Wc_loc (Kc, Ng), intercept (Ng,), sigma_log (Ng,)); `cols`: gene slice of a larger shard."""
    if hasattr(obj, "read"):
        from brie_amd import _capi
        W, b, lam = obj.read(_capi.WC_LOC), obj.read(_capi.INTERCEPT), obj.read(_capi.SIGMA_LOG)
    else:
        W, b, lam = obj.Wc_loc, obj.intercept, obj.sigma_log
    W, b, lam = np.asarray(W, np.float64), np.asarray(b, np.float64).reshape(-1), np.asarray(lam, np.float64).reshape(-1)
    if cols is not None:
        W, b, lam = W[:, cols], b[cols], lam[cols]
    return {"Wc_loc": W, "intercept": b, "sigma_log": lam}


def gene_shift(pa, pb):
    """max over a gene's own parameters of |difference| between two runs, (Ng,)."""
    w = np.abs(pa["Wc_loc"] - pb["Wc_loc"])
    w = w.max(0) if w.size else np.zeros(pa["intercept"].shape[0])
    return np.maximum(w, np.maximum(np.abs(pa["intercept"] - pb["intercept"]), np.abs(pa["sigma_log"] - pb["sigma_log"])))


def psi_parity_rule(psi, par, what=""):
    """THE parity rule of the floating-point path (north star: PSI within 1e-4 of the CPU path).
    Revision 2 (round 3).  Revision 1 was frozen on seven cases (profiles/psi_delta_r03.json) and then put to six
    held-out cases with other seeds and another shape: it held on five and failed on one
    (profiles/r3r_psi_delta_heldout.json, mid_cli_96_s3) -- which showed what it had lumped together, see "clustered".
    Revision 2 was in turn checked on a second held-out set generated after it (profiles/psi_delta.py::HELD_OUT_2).

    psi[k], par[k] for k in 'hip' (the HIP path), 'o32' (the CPU restatement in fp32 = the reference's precision),
    'o64' (the same in fp64 = the precision-independent answer); same init, same noise stream.

    Why not "every entry within 1e-4": Keras Adam moves a parameter by lr * m / (sqrt(v) + eps); the first step of each of
    the six fresh optimisers is +-lr whatever |g| is, so where a gradient is ~0 the SIGN of an fp32 rounding error decides
    an O(lr) move.  Any two fp32 evaluations of the reference's arithmetic differ by more than 1e-4 somewhere.  The evidence
    says where: when the sign event hits one of a gene's OWN parameters (a Wc_loc entry, its intercept, its sigma) all
    Nc cells of that gene move together -- at configs[2] ONE such gene of 512 carries 34 884 of the HIP path's 35 210
    entries beyond 1e-4, and the fp32 oracle has two other such genes -- and everywhere else the two fp32 runs have the
    same handful of scattered entries beyond 1e-4.  Such gene-level events hit either fp32 run with the same frequency
    (20 cases: 485 displaced + 25 clustered genes in the HIP runs, 477 + 20 in the fp32 oracle's), but each carries hundreds of entries, so they are
    counted as GENES, and only what is left is counted as ENTRIES -- each against what the reference's own precision does
    on the same trajectory:

      displaced gene: own-parameter shift vs the fp64 run > GENE_SHIFT = 4e-4 (moves Psi by 1e-4 where sigmoid' = 1/4);
      clustered gene: not displaced at the end of the fit, yet more than max(5, 0.1 % of its cells) beyond 1e-4: its cells
                      moved TOGETHER -- mostly own parameters just under the threshold (2.4e-4 .. 4e-4) acting through
                      the covariates, sometimes no end-of-fit shift at all: under noisy MC gradients the fp32 and the
                      fp64 trajectory of a gene part and re-converge again and again, and the last step is a snapshot
                      of that process (profiles/r3u_cluster_trajectory_*.json: 10 682 of 20 000 cells at step 1715,
                      0 at step 4949, 190 at the end);
      moved gene = displaced or clustered;  quiet gene = moved in NEITHER run.
      1. genes:    #moved(hip) <= 1.5 #moved(o32) + max(3, 1 % of the genes);
      2. entries of quiet genes:  #(d > 1e-4) <= 1.5 #(d32 > 1e-4) + max(1e-5 n, 50);
      3. their bulk:   p99(d) <= 1.5 p99(d32) + 1e-5   (revision 1's "max(1e-4, 1.5 p99(d32))" is implied by the
                       definition of a quiet gene and was replaced; observed ratio <= 1.48 on the thirteen cases);
      4. the worst entry of every gene not displaced in either run (clustered ones included):
                       max d <= max(2e-3, 3 max d32)   (a fifth of what one flipped +-lr step of a CELL's own Z_loc can do);
      5. a displaced gene is displaced by a bounded amount: shift <= 0.15 (the sum of the six stage learning rates is 0.051;
         observed <= 0.108 on 200-cell data, <= 0.090 at 10-20k cells, <= 0.016 at 50k cells)."""
    P = {k: np.asarray(psi[k], np.float64) for k in ("hip", "o32", "o64")}
    d, d32 = np.abs(P["hip"] - P["o64"]), np.abs(P["o32"] - P["o64"])
    s_h, s_o = gene_shift(par["hip"], par["o64"]), gene_shift(par["o32"], par["o64"])
    disp_h, disp_o = s_h > GENE_SHIFT, s_o > GENE_SHIFT
    Nc, Ng = d.shape
    cluster = max(5, int(1e-3 * Nc))
    clus_h = ~disp_h & ((d > PSI_TOL).sum(0) > cluster)
    clus_o = ~disp_o & ((d32 > PSI_TOL).sum(0) > cluster)
    moved_h, moved_o = disp_h | clus_h, disp_o | clus_o
    undisp = ~(disp_h | disp_o)
    quiet = ~(moved_h | moved_o)
    rep = {"genes": Ng, "displaced_genes": {"hip": int(disp_h.sum()), "fp32_oracle": int(disp_o.sum())},
           "clustered_genes": {"hip": int(clus_h.sum()), "fp32_oracle": int(clus_o.sum()), "more_cells_beyond_1e-4_than": cluster},
           "largest_gene_shift": {"hip": float(s_h.max()), "fp32_oracle": float(s_o.max())},
           "all_entries": {"max": float(d.max()), "p99": float(np.percentile(d, 99)), "frac_gt_1e-4": float((d > PSI_TOL).mean()),
                           "fp32_oracle": {"max": float(d32.max()), "p99": float(np.percentile(d32, 99)),
                                           "frac_gt_1e-4": float((d32 > PSI_TOL).mean())}}}
    assert moved_h.sum() <= 1.5 * moved_o.sum() + max(3, 0.01 * Ng), (what, "moved genes", rep["displaced_genes"], rep["clustered_genes"])
    assert s_h.max() <= 0.15, (what, "gene shift", float(s_h.max()))

    def stats(keep):
        dk, dk32 = d[:, keep], d32[:, keep]
        return {"genes": int(keep.sum()), "entries": int(dk.size),
                "gt_1e-4": {"hip": int((dk > PSI_TOL).sum()), "fp32_oracle": int((dk32 > PSI_TOL).sum())},
                "p99": {"hip": float(np.percentile(dk, 99)), "fp32_oracle": float(np.percentile(dk32, 99))},
                "max": {"hip": float(dk.max()), "fp32_oracle": float(dk32.max())}}
    if undisp.any():
        u = rep["undisplaced_genes"] = stats(undisp)
        assert u["max"]["hip"] <= max(2e-3, 3 * u["max"]["fp32_oracle"]), (what, "max", u["max"])
    if quiet.any():
        q = rep["quiet_genes"] = stats(quiet)
        n, n32 = q["gt_1e-4"]["hip"], q["gt_1e-4"]["fp32_oracle"]
        assert n <= 1.5 * n32 + max(1e-5 * q["entries"], 50), (what, "entries beyond 1e-4 in quiet genes", n, n32, q["entries"])
        assert q["p99"]["hip"] <= 1.5 * q["p99"]["fp32_oracle"] + 1e-5, (what, "p99", q["p99"])
    return rep


def psi_parity_of(sh, o32, o64, cols=None, what=""):
    """psi_parity_rule for a device shard against the two oracles (`cols`: the oracles hold only this gene slice)."""
    from brie_amd import _capi
    psi_h = sh.read(_capi.PSI)
    if cols is not None:
        psi_h = psi_h[:, cols]
    return psi_parity_rule({"hip": psi_h, "o32": o32.Psi, "o64": o64.Psi},
                           {"hip": run_params(sh, cols), "o32": run_params(o32), "o64": run_params(o64)}, what)
