#!/usr/bin/env python
"""The direct fp32-vs-fp32 null of the parity argument (round 4).

Three runs of every case of profiles/psi_delta.py::CASES on the same problem, init and noise stream:
  o32   oracle/brie_oracle.c in fp32 (the reference's precision, operation by operation; cached by psi_delta.py)
  o32b  the same source built with -DBRIE_ORACLE_B: float Box-Muller, reversed cell order with fp32 partial sums,
        fused multiply-adds -- a second fp32 evaluation that differs the way any other fp32 implementation may
  hip   libbrie_amd.so
and two comparisons, each reduced to PER-GENE summaries (tests/util.py::gene_summaries):
  null = o32b vs o32   (CPU only:  python profiles/psi_null.py --null --cases ...   -> profiles/psi_null/<case>_null.npz)
  hip  = hip  vs o32   (GPU box:   python profiles/psi_null.py --hip  --cases ...   -> gpurun_out/psi_null/<case>_hip.npz,
                        committed as profiles/psi_null_hip/<case>_hip.npz)
--evaluate applies tests/util.py::psi_null_rule to every case that has both and writes profiles/psi_null_r04.json.
--fixture CASE:N writes tests/golden/psi_null_<case>_first<N>.npz: the o32 Psi and parameters of the first N genes
(genes are independent and the noise is keyed by the global gene index, so the slice is exact for those genes) plus
the null summaries of those genes -- what a test needs to judge a HIP run of those N genes from a fresh clone.
The oracle is the checker here, never the thing measured.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "profiles"))
import psi_delta as pd                                      # noqa: E402  (CASES, problem(), schedule(), the o32 cache)

from tests.support.null_fixture import NULL_DIR, load_summary       # noqa: E402,F401
HIP_DIR = os.path.join(ROOT, "gpurun_out", "psi_null")        # written on the GPU box, merged back by gpurun
HIP_KEPT = os.path.join(ROOT, "profiles", "psi_null_hip")      # ... and committed from there
GOLDEN = os.path.join(ROOT, "tests", "golden")
SUMMARY_KEYS = ("shift", "n_gt", "max", "hist", "Nc")


def run_o32b(case, tag=""):
    """Psi and per-gene parameters of the o32b build after the case's full schedule (cached next to the o32 cache).
    `tag`: a further draw of the null (e.g. another OpenMP thread count = another association of the per-thread sums)."""
    path = os.path.join(pd.CACHE, "%s_float32b%s.npz" % (case, tag))
    if os.path.exists(path):
        z = np.load(path)
        return {k: z[k] for k in pd.PARAMS + ("psi",)}
    from oracle.c_oracle import COracle
    P, c = pd.problem(case)
    t0 = time.time()
    o = COracle(P["counts_pc"], P["Xc"], effLen=P["effLen"], seed=pd.model_seed(case), dtype=np.float32, variant_b=True)
    for n, lr in pd.schedule(c["min_iter"]):
        o.reset_optimizer()
        o.minimize(n, lr, c["MC"])
    out = {"psi": np.asarray(o.Psi, np.float32), "Wc_loc": np.asarray(o.Wc_loc, np.float64),
           "intercept": np.asarray(o.intercept, np.float64), "sigma_log": np.asarray(o.sigma_log, np.float64)}
    os.makedirs(pd.CACHE, exist_ok=True)
    np.savez(path, seconds=time.time() - t0, threads=o.threads(), **out)
    print("o32b %s: %.1f s on %d threads" % (case, time.time() - t0, o.threads()), flush=True)
    return out


def save_summary(path, s, **extra):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savez_compressed(path, **{k: s[k] for k in SUMMARY_KEYS}, **extra)


def null_case(case, tag=""):
    from tests import util
    o32 = pd.run_oracle(case, np.float32, want_params=True)
    b = run_o32b(case, tag)
    s = util.gene_summaries(b["psi"], o32["psi"], pd.util_params(b), pd.util_params(o32))
    save_summary(os.path.join(NULL_DIR, "%s_null%s.npz" % (case, tag)), s)
    print("null %-16s moved-by-shift %d  entries > 1e-4: %d  max %.3g" % (
        case, int((s["shift"] > util.GENE_SHIFT).sum()), int(s["n_gt"].sum()), float(s["max"].max())), flush=True)


def hip_case(case):
    from brie_amd import _capi
    from tests import util
    o32 = pd.run_oracle(case, np.float32, want_params=True)        # the cache must have travelled: never computed on a GPU box
    P, c = pd.problem(case)
    sh = util.device_shard(P, c["Nc"], c["Ng"], c["Kc"], pd.model_seed(case))
    t0 = time.time()
    for n, lr in pd.schedule(c["min_iter"]):
        sh.reset_optimizer()
        sh.step(n, lr, c["MC"], trace=False)
    psi = sh.read(_capi.PSI)
    secs = time.time() - t0
    s = util.gene_summaries(psi, o32["psi"], util.run_params(sh), pd.util_params(o32))
    sh.close()
    save_summary(os.path.join(HIP_DIR, "%s_hip.npz" % case), s, seconds=secs)
    print("hip  %-16s %.1f s  moved-by-shift %d  entries > 1e-4: %d  max %.3g" % (
        case, secs, int((s["shift"] > util.GENE_SHIFT).sum()), int(s["n_gt"].sum()), float(s["max"].max())), flush=True)


def sha256(path):
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        for blk in iter(lambda: fh.read(1 << 22), b""):
            h.update(blk)
    return h.hexdigest()


def fixture(case, n):
    """tests/golden/psi_null_<case>_first<n>.npz + the sha256 / regeneration record of the full caches."""
    from tests import util
    o32 = pd.run_oracle(case, np.float32, want_params=True)
    null = load_summary(os.path.join(NULL_DIR, "%s_null.npz" % case))
    cols = np.arange(n)
    sl = util.slice_summaries(null, cols)
    out = os.path.join(GOLDEN, "psi_null_%s_first%d.npz" % (case, n))
    np.savez_compressed(out, psi_o32=np.asarray(o32["psi"][:, :n], np.float32), Wc_loc=np.asarray(o32["Wc_loc"])[:, :n],
                        intercept=np.asarray(o32["intercept"]).reshape(-1)[:n],
                        sigma_log=np.asarray(o32["sigma_log"]).reshape(-1)[:n],
                        **{"null_" + k: sl[k] for k in SUMMARY_KEYS})
    rec_path = os.path.join(GOLDEN, "psi_null_caches.json")
    rec = json.load(open(rec_path)) if os.path.exists(rec_path) else {}
    files = {}
    for suffix in ("float32", "float32b"):
        f = os.path.join(pd.CACHE, "%s_%s.npz" % (case, suffix))
        z = np.load(f)
        files[os.path.basename(f)] = {"sha256": sha256(f), "bytes": os.path.getsize(f), "seconds": round(float(z["seconds"]), 1)}
    rec[case] = {"files": files, "fixture": os.path.basename(out), "fixture_genes": n,
                 "regenerate": "python profiles/psi_delta.py --oracles-only --cases %s && python profiles/psi_null.py --null --cases %s" % (case, case)}
    with open(rec_path, "w") as fh:
        json.dump(rec, fh, indent=1, sort_keys=True)
    print("wrote", out, os.path.getsize(out) >> 10, "KiB")


def evaluate(cases, out, hip_kept=None):
    from tests import util
    hip_kept = hip_kept or HIP_KEPT
    result = {"rule": util.psi_null_rule.__doc__, "cases": {}}
    tot = {"displaced": [0, 0], "clustered": [0, 0], "quiet_entries_gt_1e-4": [0, 0]}
    for case in cases:
        fn, fh = os.path.join(NULL_DIR, "%s_null.npz" % case), os.path.join(HIP_DIR, "%s_hip.npz" % case)
        if hip_kept != HIP_KEPT or not os.path.exists(fh):      # the committed copy of a GPU call
            fh = os.path.join(hip_kept, "%s_hip.npz" % case)
        if not (os.path.exists(fn) and os.path.exists(fh)):
            continue
        c = pd.CASES[case]
        rep = util.psi_null_rule(load_summary(fh), load_summary(fn), case, check=False)
        rep["desc"] = c["desc"]
        extra = sorted(f for f in os.listdir(NULL_DIR) if f.startswith(case + "_null_") and f.endswith(".npz"))
        if extra:                   # further draws of the null: the spread of the yardstick itself (reported, not used by the rule)
            rep["further_null_draws"] = {}
            for f in extra:
                r2 = util.psi_null_rule(load_summary(fh), load_summary(os.path.join(NULL_DIR, f)), case, check=False)
                # ... and the draw itself judged the way the HIP run is: a CPU re-evaluation held against the FIRST draw
                r3 = util.psi_null_rule(load_summary(os.path.join(NULL_DIR, f)), load_summary(fn), case, check=False)
                rep["further_null_draws"][f[len(case) + 6:-4]] = {
                    "holds_against_this_draw": r2["holds"], "displaced_genes": r2["displaced_genes"]["o32b_vs_o32"],
                    "clustered_genes": r2["clustered_genes"]["o32b_vs_o32"],
                    "quiet_gt_1e-4": r2.get("quiet_genes", {}).get("gt_1e-4"),
                    "this_draw_judged_like_the_hip_run_against_the_first_draw": {
                        "holds": r3["holds"], "violated": r3.get("violated"),
                        "displaced_genes": r3["displaced_genes"], "clustered_genes": r3["clustered_genes"],
                        "quiet_gt_1e-4": r3.get("quiet_genes", {}).get("gt_1e-4"),
                        "quiet_p99": r3.get("quiet_genes", {}).get("p99_upper_bin_edge")}}
        rep["steps"], rep["MC_size"] = 6 * int(c["min_iter"] / 6), c["MC"]
        result["cases"][case] = rep
        for k, key in (("displaced", "displaced_genes"), ("clustered", "clustered_genes")):
            tot[k][0] += rep[key]["hip_vs_o32"]
            tot[k][1] += rep[key]["o32b_vs_o32"]
        if "quiet_genes" in rep:
            tot["quiet_entries_gt_1e-4"][0] += rep["quiet_genes"]["gt_1e-4"]["hip_vs_o32"]
            tot["quiet_entries_gt_1e-4"][1] += rep["quiet_genes"]["gt_1e-4"]["o32b_vs_o32"]
        q = rep.get("quiet_genes", {})
        print("%-16s holds %-5s displaced %3d / %3d  clustered %2d / %2d  quiet > 1e-4: %5d / %5d  p99 %.2e / %.2e" % (
            case, rep["holds"], rep["displaced_genes"]["hip_vs_o32"], rep["displaced_genes"]["o32b_vs_o32"],
            rep["clustered_genes"]["hip_vs_o32"], rep["clustered_genes"]["o32b_vs_o32"],
            q.get("gt_1e-4", {}).get("hip_vs_o32", 0), q.get("gt_1e-4", {}).get("o32b_vs_o32", 0),
            q.get("p99_upper_bin_edge", {}).get("hip_vs_o32", 0), q.get("p99_upper_bin_edge", {}).get("o32b_vs_o32", 0)))
    result["sums_hip_vs_o32__o32b_vs_o32"] = tot
    result["all_hold"] = all(r["holds"] for r in result["cases"].values())
    with open(out, "w") as fh:
        json.dump(result, fh, indent=1)
    print("wrote", out, "cases:", len(result["cases"]), "all hold:", result["all_hold"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default=",".join(pd.R03 + pd.HELD_OUT + pd.HELD_OUT_2))
    ap.add_argument("--null", action="store_true")
    ap.add_argument("--null-tag", default="", help="suffix of a further null draw (run with another OMP_NUM_THREADS)")
    ap.add_argument("--hip", action="store_true")
    ap.add_argument("--evaluate", action="store_true")
    ap.add_argument("--fixture", default=None, metavar="CASE:N")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "psi_null_r04.json"))
    ap.add_argument("--hip-kept", default=None, help="directory of committed HIP summaries to evaluate (default profiles/psi_null_hip; "
                    "profiles/psi_null_hip_final_library = call r4bd, the library of the round's last commits)")
    args = ap.parse_args()
    cases = [c for c in args.cases.split(",") if c]
    if args.null:
        for case in cases:
            null_case(case, args.null_tag)
    if args.hip:
        for case in cases:
            hip_case(case)
    if args.fixture:
        case, n = args.fixture.split(":")
        fixture(case, int(n))
    if args.evaluate:
        evaluate(cases, args.out, args.hip_kept)


if __name__ == "__main__":
    main()
