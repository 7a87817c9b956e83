#!/usr/bin/env python
"""The pre-registered NULL ENSEMBLE of the fit-level parity rule (round 5).

The brie-quant default schedule (bin/quant.py:173-177: 4 998 Adam steps, MC_size 3) on gene samples of configs[1] /
configs[2] over ALL cells.  Every case is run, on the same problem, init and noise stream, through
  o32      oracle/brie_oracle.c in fp32 (the reference's precision, operation by operation)
  members  further fp32 CPU evaluations of the same algorithm (MEMBERS below: the o32b build of that source with the
           cells cut into 2 / 4 / 6 / 8 / 12 parts = "OpenMP thread counts", and one member with the exact noise stream)
  hip      libbrie_amd.so (GPU box; tests/test_gpu_fullsize.py::test_psi_ensemble_rule_...)
and each run is reduced to per-gene summaries against o32 (tests/util.py::gene_summaries).  tests/util.py::
psi_ensemble_rule holds the HIP summaries against the ensemble's and prints the ensemble's own leave-one-out record.

Order of events, kept in tests/golden/psi_ensemble_manifest.json:
  --register   writes the rule's constants, the member definitions and the cases (shapes, seeds) -- committed BEFORE the
               members of the new cases were computed and before HIP ran on the held-out cases (*_s5);
  --run        computes o32 / the members on the CPU (hours; profiles/_psi_cache/ens/, git-ignored);
  --freeze     writes tests/golden/psi_ens_<case>_first64.npz (o32 Psi + parameters, every member's summaries) and
               records the sha256 of every file in the manifest.
The oracle is the checker here, never the thing measured.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "profiles")):
    if p not in sys.path:
        sys.path.insert(0, p)
import psi_delta as pd                                      # noqa: E402

ENS = os.path.join(pd.CACHE, "ens")
from tests.support.ensemble import GOLDEN, MANIFEST       # noqa: E402
from tests.support.ensemble import (ADDENDUM_1, ADDENDUM_2, ADDENDUM_3, CASES, GENES, MEMBERS, OLD_DRAWS,   # noqa: E402,F401
                                    REGISTERED_FIRST, SLICED, load_fixture, problem, sha256)


def path_of(case, run):
    return os.path.join(ENS, "%s_%s.npz" % (case, run))


def _slice_old(case, run):
    """Round 4's caches hold 128 genes of c2_cli_128 / c3_cli_128: o32 and the draws at 4 / 6 / 8 threads."""
    if case not in SLICED:
        return None
    if run == "o32":
        src = os.path.join(pd.CACHE, "%s_float32.npz" % case)
    elif run in OLD_DRAWS:
        src = os.path.join(pd.CACHE, "%s_float32b%s.npz" % (case, OLD_DRAWS[run]))
    else:
        return None
    if not os.path.exists(src):
        return None
    z = np.load(src)
    if run != "o32" and int(z["threads"]) != MEMBERS[run]["parts"]:
        return None
    out = {"psi": np.ascontiguousarray(z["psi"][:, :GENES]), "Wc_loc": np.ascontiguousarray(z["Wc_loc"][:, :GENES]),
           "intercept": np.asarray(z["intercept"]).reshape(-1)[:GENES], "sigma_log": np.asarray(z["sigma_log"]).reshape(-1)[:GENES]}
    np.savez(path_of(case, run), seconds=float(z["seconds"]), origin="first %d genes of %s (sha256 %s)" % (
        GENES, os.path.basename(src), sha256(src)), **out)
    return out


def run_one(case, run, threads):
    """o32 or one member of `case` on the CPU -> profiles/_psi_cache/ens/<case>_<run>.npz."""
    os.makedirs(ENS, exist_ok=True)
    out = path_of(case, run)
    if os.path.exists(out):
        return
    if _slice_old(case, run) is not None:
        print("%s %s: sliced from round 4's cache" % (case, run), flush=True)
        return
    from oracle.c_oracle import COracle
    P, c, n = problem(case)
    t0 = time.time()
    o = COracle(P["counts_pc"], P["Xc"], effLen=P["effLen"], seed=pd.model_seed(CASES[case]["of"]), dtype=np.float32,
                variant_b=run != "o32")
    o.set_threads(threads)
    o.set_parts(8)                      # o32: fp64 per-part sums, cut as on the 8 threads round 4's o32 caches were made with
    if run != "o32":
        m = MEMBERS[run]
        o.set_parts(m["parts"])
        o.b_config(m["float_noise"], m["reverse"], m["chunk"])
    for k, lr in pd.schedule(c["min_iter"]):
        o.reset_optimizer()
        o.minimize(k, lr, c["MC"])
    res = {"psi": np.asarray(o.Psi, np.float32), "Wc_loc": np.asarray(o.Wc_loc, np.float64),
           "intercept": np.asarray(o.intercept, np.float64), "sigma_log": np.asarray(o.sigma_log, np.float64)}
    np.savez(out, seconds=time.time() - t0, threads=threads, origin="computed", **res)
    print("%s %s: %.0f s on %d threads" % (case, run, time.time() - t0, threads), flush=True)


def run_all(cases, runs, cores):
    """Every missing (case, run) as its own process, at most `cores` threads in flight."""
    jobs = [(c, r) for c in cases for r in runs if not os.path.exists(path_of(c, r))]
    jobs.sort(key=lambda j: (not j[0].startswith(("c5", "c3")), j[1] != "o32"))        # long ones first
    live = []
    while jobs or live:
        live = [(p, t) for p, t in live if p.poll() is None]
        used = sum(t for _, t in live)
        started = False
        for j in list(jobs):
            parts = 8 if j[1] == "o32" else MEMBERS[j[1]]["parts"]
            t = next(d for d in (4, 3, 2, 1) if parts % d == 0 and d <= cores)   # threads divide the parts: no idle rounds
            if used + t <= cores:
                env = dict(os.environ, OMP_NUM_THREADS=str(t), OMP_WAIT_POLICY="active")
                live.append((subprocess.Popen([sys.executable, os.path.abspath(__file__), "--one", "%s:%s:%d" % (j[0], j[1], t)],
                                              env=env), t))
                jobs.remove(j)
                used += t
                started = True
        if not started:
            time.sleep(5)


def load(case, run):
    z = np.load(path_of(case, run))
    return {k: z[k] for k in pd.PARAMS + ("psi",)}


def register():
    from tests import util
    reg = {"rule": "tests/util.py::psi_ensemble_rule",
           "constants": {"factor": util.ENSEMBLE_FACTOR, "max_floor": util.ENSEMBLE_MAX_FLOOR, "max_factor": util.ENSEMBLE_MAX_FACTOR,
                         "shift_cap": util.ENSEMBLE_SHIFT_CAP, "floors": "moved_genes max(2, 1 % genes); quiet_rate max(1e-5, 20 / entries); "
                         "quiet_p99 5e-6", "psi_tol": util.PSI_TOL, "gene_shift": util.GENE_SHIFT},
           "genes": GENES, "members": MEMBERS, "member_build": "gcc -O3 -DBRIE_ORACLE_B -ffp-contract=fast -mfma -fopenmp oracle/brie_oracle.c",
           "cases": {k: dict({kk: vv for kk, vv in pd.CASES[CASES[k]["of"]].items()}, model_seed=pd.model_seed(CASES[k]["of"]),
                             held_out=bool(CASES[k].get("held_out"))) for k in REGISTERED_FIRST},
           "note": "registered before the members of the *_s5 cases (and t2 / t12 / x3 of the others) were computed and before "
                   "the HIP path ran on the *_s5 cases; t4 / t6 / t8 of c2_cli_128 / c3_cli_128 are round 4's draws"}
    man = json.load(open(MANIFEST)) if os.path.exists(MANIFEST) else {}
    if "registered" in man and man["registered"] != json.loads(json.dumps(reg)):
        raise SystemExit("the manifest is already registered with other contents: not overwritten")
    man["registered"] = reg
    add = {"cases": {k: dict({kk: vv for kk, vv in pd.CASES[v["of"]].items()}, model_seed=pd.model_seed(v["of"]), held_out=True)
                     for k, v in ADDENDUM_1.items()},
           "note": "same rule, constants and members as `registered`; added after the four cases of `registered` had been judged on the "
                   "GPU (call r5q: all hold) and before any run of either side on these three"}
    if "registered_addendum_1" in man and man["registered_addendum_1"] != json.loads(json.dumps(add)):
        raise SystemExit("addendum 1 is already registered with other contents: not overwritten")
    man["registered_addendum_1"] = add
    add2 = {"cases": {k: dict({kk: vv for kk, vv in pd.CASES[v["of"]].items()}, model_seed=pd.model_seed(v["of"]),
                              held_out=bool(v.get("held_out"))) for k, v in ADDENDUM_2.items()},
            "note": "the API schedule (996 steps, MC_size 1) under the same rule, constants and members; registered after call r5zz; "
                    "c2_api_512 / c3_api_512 / c3_api_512_s2 are round 4's gene-sample cases (HIP has run on them many times: in "
                    "sample), the *_s7 cases are held out"}
    if "registered_addendum_2" in man and man["registered_addendum_2"] != json.loads(json.dumps(add2)):
        raise SystemExit("addendum 2 is already registered with other contents: not overwritten")
    man["registered_addendum_2"] = add2
    add3 = {"cases": {k: dict({kk: vv for kk, vv in pd.CASES[v["of"]].items()}, model_seed=pd.model_seed(v["of"]), held_out=True)
                      for k, v in ADDENDUM_3.items()},
            "note": "same rule, constants and members; registered after the twelve cases above had been judged (all hold) and before "
                    "any run of either side on these two: the BASELINE shapes no case had (configs[4]: 100 000 cells, 2 layers, "
                    "Kc = 5; configs[0]: 200 cells, Kc = 0), brie-quant default schedule, first 64 genes"}
    if "registered_addendum_3" in man and man["registered_addendum_3"] != json.loads(json.dumps(add3)):
        raise SystemExit("addendum 3 is already registered with other contents: not overwritten")
    man["registered_addendum_3"] = add3
    with open(MANIFEST, "w") as fh:
        json.dump(man, fh, indent=1, sort_keys=True)
    print("registered", MANIFEST)


def freeze(cases):
    from tests import util
    man = json.load(open(MANIFEST))
    man.setdefault("frozen", {})
    for case in cases:
        if not all(os.path.exists(path_of(case, r)) for r in ["o32"] + list(MEMBERS)):
            print("freeze: %s incomplete, skipped" % case)
            continue
        o32 = load(case, "o32")
        blob = {"psi_o32": np.asarray(o32["psi"], np.float32), "Wc_loc": o32["Wc_loc"], "intercept": o32["intercept"],
                "sigma_log": o32["sigma_log"], "Nc": o32["psi"].shape[0]}
        base = os.path.join(GOLDEN, "psi_null_%s_first%d.npz" % (case, GENES))
        if os.path.exists(base):            # round 4's fixture of the same genes already holds the o32 run: not stored twice
            zb = np.load(base)
            assert np.array_equal(zb["psi_o32"], blob["psi_o32"]) and np.array_equal(zb["Wc_loc"], blob["Wc_loc"]), case
            blob = {"Nc": blob["Nc"], "o32_in": os.path.basename(base)}
        files, summ = {"o32": {"sha256": sha256(path_of(case, "o32"))}}, {}
        for m in sorted(MEMBERS):
            r = load(case, m)
            s = util.gene_summaries(r["psi"], o32["psi"], pd.util_params(r), pd.util_params(o32))
            summ[m] = s
            for k in ("shift", "n_gt", "max", "hist"):
                blob["%s_%s" % (m, k)] = s[k]
            z = np.load(path_of(case, m))
            files[m] = {"sha256": sha256(path_of(case, m)), "seconds": round(float(z["seconds"]), 1), "origin": str(z["origin"])}
        out = os.path.join(GOLDEN, "psi_ens_%s_first%d.npz" % (case, GENES))
        np.savez_compressed(out, **blob)
        # (+ the ensemble's own record under the rule: leave-one-out, HIP not involved)
        man["frozen"][case] = {"files": files, "fixture": os.path.basename(out), "fixture_sha256": sha256(out),
                               "leave_one_out_of_all_members": _loo(summ)}
        print("froze %s -> %s (%d KiB); leave-one-out failures: %s" % (case, out, os.path.getsize(out) >> 10,
                                                                        man["frozen"][case]["leave_one_out_of_all_members"]))
    with open(MANIFEST, "w") as fh:
        json.dump(man, fh, indent=1, sort_keys=True)


def _loo(summ):
    from tests import util
    ms = {k: util.comparison_stats(v) for k, v in summ.items()}
    out = {}
    for k in sorted(ms):
        v = util._ensemble_violations(ms[k], [ms[j] for j in ms if j != k])
        out[k] = [list(t) for t in v]
    return {"failing": sorted(k for k, v in out.items() if v), "of": len(out), "violations": {k: v for k, v in out.items() if v}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default=",".join(CASES))      # (run / freeze skip what exists)
    ap.add_argument("--runs", default=",".join(["o32"] + list(MEMBERS)))
    ap.add_argument("--register", action="store_true")
    ap.add_argument("--run", action="store_true")
    ap.add_argument("--cores", type=int, default=6)
    ap.add_argument("--one", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--freeze", action="store_true")
    args = ap.parse_args()
    if args.one:
        case, run, t = args.one.split(":")
        return run_one(case, run, int(t))
    cases = [c for c in args.cases.split(",") if c]
    if args.register:
        register()
    if args.run:
        run_all(cases, [r for r in args.runs.split(",") if r], args.cores)
    if args.freeze:
        freeze(cases)


if __name__ == "__main__":
    main()
