#!/usr/bin/env python
"""The cases of the fit-level parity evidence (gene samples of the BASELINE configs over ALL cells, both default schedules)
and their CPU oracle runs, cached under profiles/_psi_cache (git-ignored, reproducible; tests/golden/psi_null_caches.json).

    python profiles/psi_delta.py --oracles-only --cases c3_api_512          (no GPU: fills the oracle cache)

Schedules: BRIE2.fit defaults (6 x 166 = 996 steps, MC_size 1; model_TFProb.py:214-241) and the brie-quant defaults
(6 x 833 = 4998 steps, MC_size 3; bin/quant.py:173-177).  Genes are independent (model_wrap.py:241), so a gene sample over
ALL cells of configs[1] / configs[2] is exact for those genes.  The HIP side of the comparison lives in profiles/psi_null.py
(one null draw, round 4) and profiles/psi_ensemble.py (the pre-registered ensemble, round 5); rounds 2 - 3 compared three
HIP builds (strict Adam, strict transcendentals) against the fp64 oracle from here -- profiles/history/psi_delta_r0{2,3}.json.
The oracle is the checker here, never the thing measured.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from tests.support.psi_cases import (CACHE, CASES, HELD_OUT, HELD_OUT_2, PARAMS, QUICK, R03, SEED, model_seed,   # noqa: E402,F401
                                     problem, schedule, util_params)


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default=",".join(R03))
    ap.add_argument("--oracles-only", action="store_true", help="fill profiles/_psi_cache with the fp32 and fp64 oracle runs of the cases")
    ap.add_argument("--slice-cache", default=None, metavar="CASE:N",
                    help="write the first N genes of a case's oracle cache as <case>_firstN_<dtype>.npz (genes are independent)")
    args = ap.parse_args()
    if args.slice_cache:
        case, n = args.slice_cache.split(":")
        n = int(n)
        for dt in ("float32", "float64"):
            z = np.load(os.path.join(CACHE, "%s_%s.npz" % (case, dt)))
            np.savez(os.path.join(CACHE, "%s_first%d_%s.npz" % (case, n, dt)), psi=z["psi"][:, :n], Wc_loc=z["Wc_loc"][:, :n],
                     intercept=np.asarray(z["intercept"]).reshape(-1)[:n], sigma_log=np.asarray(z["sigma_log"]).reshape(-1)[:n],
                     seconds=z["seconds"])
        return
    for case in [c for c in args.cases.split(",") if c]:
        run_oracle(case, np.float32, want_params=True)
        run_oracle(case, np.float64, want_params=True)


if __name__ == "__main__":
    main()
