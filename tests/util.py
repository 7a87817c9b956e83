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
                 mode='gene', variant_b=False):
    o = OracleBRIE2(Nc, Ng, Kc, effLen=P["effLen"], seed=seed, dtype=dtype,
                    gene_offset=gene_offset, intercept=intercept, sigma=sigma, Kg=Kg, intercept_mode=mode,
                    variant_b=variant_b)
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


def states_close_stats(so, sd, worst=1e-3):
    """Per state array: (elements, 99.9 % quantile of |difference|, largest, elements beyond `worst`)."""
    out = {}
    for k in STATE_KEYS:
        if k not in so or np.asarray(so[k]).size == 0:
            continue
        d = np.abs(np.asarray(so[k], np.float64) - np.asarray(sd[k], np.float64))
        out[k] = (int(d.size), float(np.percentile(d, 99.9)), float(d.max()), int((d >= worst).sum()))
    return out


def states_close_violations(so, sd, bulk=1e-5, worst=1e-3, lr=0.01, fresh=1):
    """THE short-horizon parity rule (tests/test_gpu_parity.py::assert_states_close states where its constants come
    from) as a list of violations, so that the negative controls (tests/test_rule_power.py) can ask whether a WRONG
    algorithm is rejected by exactly the bounds the HIP path is held to:
      bulk    99.9 % of every array of >= 1000 elements within `bulk`;
      worst   at most max(1, 1e-4 n) elements of an array beyond `worst`, none beyond max(worst, 2.2 lr fresh)."""
    viol = []
    for k, (n, p999, mx, n_out) in states_close_stats(so, sd, worst).items():
        if n >= 1000 and not p999 < bulk:
            viol.append((k, "99.9 % quantile", p999))
        if not n_out <= max(1, int(1e-4 * n)):
            viol.append((k, "elements beyond %g" % worst, n_out, mx))
        if not mx < max(worst, 2.2 * lr * fresh):
            viol.append((k, "max", mx))
    return viol


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
    Revision 2 (round 3).  Revision 1 was frozen on seven cases (profiles/history/psi_delta_r03.json) and then put to six
    held-out cases with other seeds and another shape: it held on five and failed on one
    (profiles/history/r3r_psi_delta_heldout.json, mid_cli_96_s3) -- which showed what it had lumped together, see "clustered".
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
                      of that process (profiles/history/r3u_cluster_trajectory_*.json: 10 682 of 20 000 cells at step 1715,
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


# ---- the direct fp32-vs-fp32 null (round 4) ----------------------------------------------------------
NULL_BINS = 10.0 ** (-8.0 + 0.02 * np.arange(401))       # log-spaced edges 1e-8 .. 1, 4.7 % wide


def gene_summaries(psi_a, psi_b, par_a, par_b):
    """Everything the null rule consumes about the difference of two runs, PER GENE (so that it can be computed where
    the matrices are and judged elsewhere): own-parameter shift, entries beyond PSI_TOL, largest entry, and a
    log-binned histogram of |dPsi| (NULL_BINS; bin 0 also takes everything below 1e-8)."""
    d = np.abs(np.asarray(psi_a, np.float64) - np.asarray(psi_b, np.float64))
    Nc, Ng = d.shape
    B = NULL_BINS.size
    hist = np.zeros((Ng, B), np.int32)
    for j0 in range(0, Ng, 64):                            # gene slabs keep the temporaries small
        dj = d[:, j0:j0 + 64]
        idx = np.clip(np.searchsorted(NULL_BINS, dj.ravel(), side="right") - 1, 0, B - 1)
        key = np.tile(np.arange(dj.shape[1]), dj.shape[0]) * B + idx
        hist[j0:j0 + dj.shape[1]] = np.bincount(key, minlength=dj.shape[1] * B).reshape(dj.shape[1], B)
    return {"shift": gene_shift(par_a, par_b), "n_gt": (d > PSI_TOL).sum(0).astype(np.int64), "max": d.max(0),
            "hist": hist, "Nc": int(Nc)}


def slice_summaries(s, cols):
    return {"shift": s["shift"][cols], "n_gt": s["n_gt"][cols], "max": s["max"][cols], "hist": s["hist"][cols], "Nc": s["Nc"]}


def _p99_from_hist(hist):
    tot = hist.sum(0).astype(np.int64)
    n = int(tot.sum())
    if n == 0:
        return 0.0
    k = int(np.searchsorted(np.cumsum(tot), 0.99 * n))
    k = min(k, NULL_BINS.size - 1)
    return float(NULL_BINS[min(k + 1, NULL_BINS.size - 1)])         # upper edge of the bin holding the 99th percentile


def psi_null_rule(h, n, what="", check=True):
    """THE parity rule since round 4: the HIP path against the fp32 CPU restatement (o32), judged by what a SECOND fp32
    CPU evaluation of the same algorithm (o32b: oracle/brie_oracle.c built with -DBRIE_ORACLE_B -- float Box-Muller,
    reversed cell order with fp32 partial sums, fused multiply-adds) does against that same o32 run.  No fp64 run is
    involved.  `h` = gene_summaries(HIP, o32), `n` = gene_summaries(o32b, o32): same problem, init and noise stream.

    The statistics and the constants are those of revision 2 (round 3, psi_parity_rule above), unchanged; only the
    yardstick changed from "o32 vs o64" to "o32b vs o32" -- a direct fp32-vs-fp32 null:
      displaced gene: own-parameter shift vs o32 > GENE_SHIFT;  clustered gene: not displaced, more than
      max(5, 0.1 % of its cells) beyond PSI_TOL;  moved = either;  quiet = moved in NEITHER comparison.
      1. #moved(h) <= 1.5 #moved(n) + max(3, 1 % of the genes);
      2. entries of quiet genes beyond 1e-4:  N(h) <= 1.5 N(n) + max(1e-5 entries, 50);
      3. their 99th percentile (from the per-gene histograms, upper bin edge): p99(h) <= 1.5 p99(n) + 1e-5;
      4. worst entry of every gene displaced in neither: max(h) <= max(2e-3, 3 max(n));
      5. largest own-parameter shift <= 0.15.
    Returns the report; with check=True a violated rule raises AssertionError naming the case (`what`)."""
    Ng, Nc = h["shift"].shape[0], h["Nc"]
    cluster = max(5, int(1e-3 * Nc))
    disp_h, disp_n = h["shift"] > GENE_SHIFT, n["shift"] > GENE_SHIFT
    clus_h, clus_n = ~disp_h & (h["n_gt"] > cluster), ~disp_n & (n["n_gt"] > cluster)
    moved_h, moved_n = disp_h | clus_h, disp_n | clus_n
    undisp, quiet = ~(disp_h | disp_n), ~(moved_h | moved_n)
    rep = {"genes": int(Ng), "cells": int(Nc),
           "displaced_genes": {"hip_vs_o32": int(disp_h.sum()), "o32b_vs_o32": int(disp_n.sum())},
           "clustered_genes": {"hip_vs_o32": int(clus_h.sum()), "o32b_vs_o32": int(clus_n.sum()),
                               "more_cells_beyond_1e-4_than": cluster},
           "largest_gene_shift": {"hip_vs_o32": float(h["shift"].max()), "o32b_vs_o32": float(n["shift"].max())},
           "all_entries_gt_1e-4": {"hip_vs_o32": int(h["n_gt"].sum()), "o32b_vs_o32": int(n["n_gt"].sum())},
           "all_entries_max": {"hip_vs_o32": float(h["max"].max()), "o32b_vs_o32": float(n["max"].max())}}
    # reported, not judged: how far the entries of the DISPLACED genes go in either comparison -- a displaced
    # gene moves all its cells by about sigmoid'(z) x its parameter shift, which rule 5 bounds
    rep["displaced_genes_worst_entry"] = {"hip_vs_o32": float(h["max"][disp_h].max()) if disp_h.any() else 0.0,
                                          "o32b_vs_o32": float(n["max"][disp_n].max()) if disp_n.any() else 0.0}
    viol = []
    if not moved_h.sum() <= 1.5 * moved_n.sum() + max(3, 0.01 * Ng):
        viol.append(("moved genes", int(moved_h.sum()), int(moved_n.sum())))
    if not h["shift"].max() <= 0.15:
        viol.append(("gene shift", float(h["shift"].max())))
    if undisp.any():
        mh, mn = float(h["max"][undisp].max()), float(n["max"][undisp].max())
        rep["undisplaced_genes"] = {"genes": int(undisp.sum()), "max": {"hip_vs_o32": mh, "o32b_vs_o32": mn}}
        if not mh <= max(2e-3, 3 * mn):
            viol.append(("max over undisplaced genes", mh, mn))
    if quiet.any():
        entries = int(quiet.sum()) * Nc
        nh, nn = int(h["n_gt"][quiet].sum()), int(n["n_gt"][quiet].sum())
        ph, pn = _p99_from_hist(h["hist"][quiet]), _p99_from_hist(n["hist"][quiet])
        rep["quiet_genes"] = {"genes": int(quiet.sum()), "entries": entries, "gt_1e-4": {"hip_vs_o32": nh, "o32b_vs_o32": nn},
                              "p99_upper_bin_edge": {"hip_vs_o32": ph, "o32b_vs_o32": pn}}
        if not nh <= 1.5 * nn + max(1e-5 * entries, 50):
            viol.append(("entries beyond 1e-4 in quiet genes", nh, nn, entries))
        if not ph <= 1.5 * pn + 1e-5:
            viol.append(("p99 of quiet genes", ph, pn))
    rep["holds"] = not viol
    if viol:
        rep["violated"] = [list(v) for v in viol]
    if check:
        assert not viol, (what, viol, rep)
    return rep


def psi_null_of(sh, o32, o32b, cols=None, what=""):
    """psi_null_rule for a device shard: HIP vs the fp32 oracle, judged by a second fp32 CPU evaluation vs that oracle
    (`cols`: the oracles hold only this gene slice of the shard)."""
    from brie_amd import _capi
    psi_h = sh.read(_capi.PSI)
    if cols is not None:
        psi_h = psi_h[:, cols]
    h = gene_summaries(psi_h, o32.Psi, run_params(sh, cols), run_params(o32))
    n = gene_summaries(o32b.Psi, o32.Psi, run_params(o32b), run_params(o32))
    return psi_null_rule(h, n, what)


# ---- the pre-registered null ENSEMBLE (round 5) -------------------------------------------------------
# One null draw is a yardstick noisier than the 1.5x margin of psi_null_rule (three draws of c3_cli_128 left 118-131 /
# 202-207 / 284-313 scattered entries, docs/evidence_r4.md section 3), so the fit-level rule for the brie-quant default
# schedule holds the HIP run against an ENSEMBLE of fp32 CPU evaluations, fixed in tests/golden/psi_ensemble_manifest.json
# (members, cases, seeds, constants) BEFORE any member of the new cases was computed and before HIP ran on the held-out ones.
ENSEMBLE_FACTOR = 1.25          # each statistic of the judged run <= FACTOR x the ensemble's largest + its floor
ENSEMBLE_FLOORS = {"moved_genes": lambda Ng, entries: max(2.0, 0.01 * Ng),        # genes
                   "quiet_rate": lambda Ng, entries: max(1e-5, 20.0 / max(entries, 1)),   # entries beyond 1e-4 per entry
                   "quiet_p99": lambda Ng, entries: 5e-6}                         # Psi units
ENSEMBLE_MAX_FLOOR, ENSEMBLE_MAX_FACTOR = 2e-3, 1.5      # worst entry of the genes not displaced: <= max(FLOOR, FACTOR x ensemble)
ENSEMBLE_SHIFT_CAP = 0.15       # largest own-parameter shift, absolute (as psi_null_rule's rule 5)


def comparison_stats(s):
    """The statistics of ONE comparison (a run vs the fp32 oracle o32; `s` = gene_summaries(run, o32)) that
    psi_ensemble_rule judges.  Self-contained: the partition into displaced / clustered / quiet genes is that of the
    comparison itself (psi_null_rule's pairwise 'moved in neither' needs a partner; an ensemble has several)."""
    Ng, Nc = s["shift"].shape[0], s["Nc"]
    cluster = max(5, int(1e-3 * Nc))
    disp = s["shift"] > GENE_SHIFT
    clus = ~disp & (s["n_gt"] > cluster)
    quiet = ~(disp | clus)
    entries = int(quiet.sum()) * Nc
    n_q = int(s["n_gt"][quiet].sum())
    return {"genes": int(Ng), "cells": int(Nc), "displaced_genes": int(disp.sum()), "clustered_genes": int(clus.sum()),
            "moved_genes": int((disp | clus).sum()),
            "quiet_entries": entries, "quiet_gt_1e-4": n_q, "quiet_rate": n_q / max(entries, 1),
            "quiet_p99": _p99_from_hist(s["hist"][quiet]) if quiet.any() else 0.0,
            "undisplaced_max": float(s["max"][~disp].max()) if (~disp).any() else 0.0,
            "shift_max": float(s["shift"].max()), "all_gt_1e-4": int(s["n_gt"].sum()), "all_max": float(s["max"].max())}


def _ensemble_violations(x, members):
    """Violations of the ensemble rule by the statistics `x` against the list of member statistics."""
    viol = []
    Ng = x["genes"]
    entries = max([x["quiet_entries"]] + [m["quiet_entries"] for m in members])
    for k in ("moved_genes", "quiet_rate", "quiet_p99"):
        top = max(m[k] for m in members)
        bound = ENSEMBLE_FACTOR * top + ENSEMBLE_FLOORS[k](Ng, entries)
        if not x[k] <= bound:
            viol.append((k, x[k], top, bound))
    top = max(m["undisplaced_max"] for m in members)
    bound = max(ENSEMBLE_MAX_FLOOR, ENSEMBLE_MAX_FACTOR * top)
    if not x["undisplaced_max"] <= bound:
        viol.append(("undisplaced_max", x["undisplaced_max"], top, bound))
    if not x["shift_max"] <= ENSEMBLE_SHIFT_CAP:
        viol.append(("shift_max", x["shift_max"], ENSEMBLE_SHIFT_CAP, ENSEMBLE_SHIFT_CAP))
    return viol


def psi_ensemble_rule(h, members, what="", check=True):
    """THE fit-level parity rule since round 5 for the brie-quant default schedule (4 998 steps, MC_size 3): the HIP path
    against the fp32 CPU restatement o32, held against an ENSEMBLE of further fp32 CPU evaluations of the same algorithm
    against that same o32 run (same problem, init and noise stream).  `h` = gene_summaries(HIP, o32); `members` = dict
    name -> gene_summaries(member, o32).  The members are fixed by tests/golden/psi_ensemble_manifest.json (o32b at 2 / 4 /
    6 / 8 / 12 OpenMP threads + one member with the EXACT noise stream, forward order and 128-cell partial sums), NOT chosen
    after seeing a HIP run.  Per comparison (comparison_stats): displaced gene = own-parameter shift > 4e-4; clustered =
    not displaced, more than max(5, 0.1 % of its cells) beyond 1e-4; quiet = neither.  With E_k the LARGEST value of
    statistic k over the members:
      1. moved genes (displaced + clustered)        <= 1.25 E_1 + max(2, 1 % of the genes)
      2. entries of quiet genes beyond 1e-4, per entry  <= 1.25 E_2 + max(1e-5, 20 / entries)
      3. 99th percentile over the quiet genes        <= 1.25 E_3 + 5e-6
      4. worst entry of the genes not displaced      <= max(2e-3, 1.5 E_4)
      5. largest own-parameter shift                 <= 0.15
    The report also carries the LEAVE-ONE-OUT record of the ensemble itself: every member judged by the same rule against
    the other members -- the rule's own false-alarm rate on runs that are the reference's arithmetic by construction.
    A case that fails is reported as failing; the constants above are not revisited."""
    names = sorted(members)
    ms = {k: comparison_stats(members[k]) for k in names}
    x = comparison_stats(h)
    viol = _ensemble_violations(x, [ms[k] for k in names])
    loo = {}
    for k in names:
        others = [ms[j] for j in names if j != k]
        if others:
            v = _ensemble_violations(ms[k], others)
            loo[k] = [list(t) for t in v]
    keys = ("moved_genes", "displaced_genes", "clustered_genes", "quiet_gt_1e-4", "quiet_rate", "quiet_p99",
            "undisplaced_max", "shift_max")
    rep = {"genes": x["genes"], "cells": x["cells"], "hip_vs_o32": {k: x[k] for k in keys},
           "ensemble_vs_o32": {n: {k: ms[n][k] for k in keys} for n in names},
           "leave_one_out": {"members_failing": sorted(k for k, v in loo.items() if v), "of": len(loo),
                             "violations": {k: v for k, v in loo.items() if v}},
           "holds": not viol}
    if viol:
        rep["violated"] = [list(v) for v in viol]
    if check:
        assert not viol, (what, viol, rep)
    return rep


def entry_ensemble_rule(h, members, n, what="", check=True):
    """The ensemble rule for the COUPLED model variants (gene features with per-cell weights, per-cell intercept, ...): one sign
    event in a parameter shared by a row reaches every gene of that cell, so the gene-level partition of psi_ensemble_rule has
    no meaning and its ENTRY-level statistics are applied to the whole matrix, with the same constants: over the `n` entries,
    `h` and every member a dict with 'n_gt_1e-4', 'p99', 'max' of |dPsi| against the fp32 oracle;
      entries beyond 1e-4, per entry  <= 1.25 E + max(1e-5, 20 / n);   p99 <= 1.25 E + 5e-6;   max <= max(2e-3, 1.5 E)
    with E the largest value over the members, and the members' own leave-one-out record beside it."""
    def viol_of(x, ms):
        v = []
        top = max(m["n_gt_1e-4"] for m in ms) / float(n)
        if not x["n_gt_1e-4"] / float(n) <= ENSEMBLE_FACTOR * top + ENSEMBLE_FLOORS["quiet_rate"](0, n):
            v.append(("rate", x["n_gt_1e-4"] / float(n), top))
        top = max(m["p99"] for m in ms)
        if not x["p99"] <= ENSEMBLE_FACTOR * top + ENSEMBLE_FLOORS["quiet_p99"](0, n):
            v.append(("p99", x["p99"], top))
        top = max(m["max"] for m in ms)
        if not x["max"] <= max(ENSEMBLE_MAX_FLOOR, ENSEMBLE_MAX_FACTOR * top):
            v.append(("max", x["max"], top))
        return v
    names = sorted(members)
    viol = viol_of(h, [members[k] for k in names])
    loo = {k: viol_of(members[k], [members[j] for j in names if j != k]) for k in names}
    rep = {"entries": int(n), "hip_vs_o32": {k: h[k] for k in ("n_gt_1e-4", "p99", "max")},
           "ensemble_vs_o32": {k: {q: members[k][q] for q in ("n_gt_1e-4", "p99", "max")} for k in names},
           "leave_one_out": {"members_failing": sorted(k for k, v in loo.items() if v), "of": len(names)}, "holds": not viol}
    if viol:
        rep["violated"] = [list(v) for v in viol]
    if check:
        assert not viol, (what, viol, rep)
    return rep
