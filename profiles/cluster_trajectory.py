#!/usr/bin/env python
"""Where does a CLUSTERED gene come from?  CPU only: the fp32 and the fp64 run of the C restatement on
the gene's quad alone (genes are independent; the noise stream is keyed by the global gene index; same thread count as
the cached runs, so the same rounding order), own-parameter shift fp32-vs-fp64 recorded every `every` steps.
    python profiles/cluster_trajectory.py mid_cli_96_s4 95 [every]
The oracle is the subject here, not a checker of anything: this script explains a property of the REFERENCE's precision."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from profiles import psi_delta as pd       # noqa: E402


def main():
    case, gene = sys.argv[1], int(sys.argv[2])
    every = int(sys.argv[3]) if len(sys.argv) > 3 else 49
    from oracle.c_oracle import COracle
    P, c = pd.problem(case)
    q0 = gene // 4 * 4
    cols = slice(q0, q0 + 4)
    cnt = [np.ascontiguousarray(x[:, cols]) for x in P["counts_pc"]]
    eff = None if P["effLen"] is None else P["effLen"][cols]
    runs = {dt: COracle(cnt, P["Xc"], effLen=eff, seed=pd.model_seed(case), gene_offset=q0, dtype=dt) for dt in (np.float32, np.float64)}
    g = gene - q0
    rows, step = [], 0
    for n, lr in pd.schedule(c["min_iter"]):
        for o in runs.values():
            o.reset_optimizer()
        done = 0
        while done < n:
            k = min(every, n - done)
            for o in runs.values():
                o.minimize(k, lr, c["MC"])
            done += k
            step += k
            a, b = runs[np.float32], runs[np.float64]
            w = np.abs(np.asarray(a.Wc_loc, np.float64)[:, g] - np.asarray(b.Wc_loc, np.float64)[:, g])
            sh = max(w.max() if w.size else 0.0, abs(float(np.ravel(a.intercept)[g]) - float(np.ravel(b.intercept)[g])),
                     abs(float(np.ravel(a.sigma_log)[g]) - float(np.ravel(b.sigma_log)[g])))
            d = np.abs(np.asarray(a.Psi, np.float64)[:, g] - np.asarray(b.Psi, np.float64)[:, g])
            rows.append({"step": step, "lr": lr, "own_parameter_shift": float(sh), "cells_beyond_1e-4": int((d > 1e-4).sum())})
    peak = max(rows, key=lambda r: r["own_parameter_shift"])
    out = {"case": case, "gene": gene, "cells": int(c["Nc"]), "every": every, "peak": peak, "end": rows[-1], "trajectory": rows}
    path = os.path.join(ROOT, "profiles", "r3u_cluster_trajectory_%s_gene%d.json" % (case, gene))
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(case, "gene", gene, "peak", peak, "end", rows[-1])
    for r in rows[::max(1, len(rows) // 24)]:
        print("  step %5d lr %.3f shift %.3g cells>1e-4 %d" % (r["step"], r["lr"], r["own_parameter_shift"], r["cells_beyond_1e-4"]))


if __name__ == "__main__":
    main()
