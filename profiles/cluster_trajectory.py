#!/usr/bin/env python
"""Where does a CLUSTERED gene come from?  CPU only: the fp32 and the fp64 run of the C restatement on
the gene's quad alone (genes are independent; the noise stream is keyed by the global gene index; same thread count as
the cached runs, so the same rounding order), own-parameter shift fp32-vs-fp64 recorded every `every` steps.
    python profiles/cluster_trajectory.py mid_cli_96_s4 95 [every] [--hip]
--hip (GPU box): the HIP path on the same quad next to the two oracle runs (a gene shard repeats the whole fit bit for
bit: tests/test_gpu_fullsize.py), its difference to the fp64 run recorded the same way.
The oracle is the subject here, not a checker of anything: this script explains a property of the REFERENCE's precision."""
import json
import os
import sys

os.environ.setdefault("OMP_WAIT_POLICY", "passive")      # 8 threads on the 6 cores of a GPU box: do not spin at barriers

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from profiles import psi_delta as pd       # noqa: E402


def main():
    hip = "--hip" in sys.argv
    argv = [a for a in sys.argv if a != "--hip"]
    case, gene = argv[1], int(argv[2])
    every = int(argv[3]) if len(argv) > 3 else 49
    from oracle.c_oracle import COracle
    P, c = pd.problem(case)
    q0 = gene // 4 * 4
    cols = slice(q0, q0 + 4)
    cnt = [np.ascontiguousarray(x[:, cols]) for x in P["counts_pc"]]
    eff = None if P["effLen"] is None else P["effLen"][cols]
    runs = {dt: COracle(cnt, P["Xc"], effLen=eff, seed=pd.model_seed(case), gene_offset=q0, dtype=dt) for dt in (np.float32, np.float64)}
    for o in runs.values():
        o.set_threads(8)          # the thread count the cached runs were written with: same rounding order of the cell sums
                                  # (and never the 256 "CPUs" a GPU box reports for the 6 cores it grants)
    g = gene - q0
    sh = None
    if hip:
        from brie_amd import _capi
        from tests import util
        Pq = dict(P, counts=[np.ascontiguousarray(x[:, cols]) for x in P["counts"]], effLen=eff)
        sh = util.device_shard(Pq, c["Nc"], 4, c["Kc"], pd.model_seed(case), gene_offset=q0)
    rows, step = [], 0
    for n, lr in pd.schedule(c["min_iter"]):
        for o in runs.values():
            o.reset_optimizer()
        if sh is not None:
            sh.reset_optimizer()
        done = 0
        while done < n:
            k = min(every, n - done)
            for o in runs.values():
                o.minimize(k, lr, c["MC"])
            if sh is not None:
                sh.step(k, lr, c["MC"], trace=False)
            done += k
            step += k
            a, b = runs[np.float32], runs[np.float64]
            w = np.abs(np.asarray(a.Wc_loc, np.float64)[:, g] - np.asarray(b.Wc_loc, np.float64)[:, g])
            sh_ = max(w.max() if w.size else 0.0, abs(float(np.ravel(a.intercept)[g]) - float(np.ravel(b.intercept)[g])),
                     abs(float(np.ravel(a.sigma_log)[g]) - float(np.ravel(b.sigma_log)[g])))
            d = np.abs(np.asarray(a.Psi, np.float64)[:, g] - np.asarray(b.Psi, np.float64)[:, g])
            rows.append({"step": step, "lr": lr, "own_parameter_shift": float(sh_), "cells_beyond_1e-4": int((d > 1e-4).sum())})
            if sh is not None:
                ph = util.run_params(sh)
                pb = util.run_params(b)
                rows[-1]["hip_own_parameter_shift"] = float(util.gene_shift(ph, pb)[g])
                dh = np.abs(sh.read(_capi.PSI)[:, g].astype(np.float64) - np.asarray(b.Psi, np.float64)[:, g])
                rows[-1]["hip_cells_beyond_1e-4"] = int((dh > 1e-4).sum())
    peak = max(rows, key=lambda r: r["own_parameter_shift"])
    out = {"case": case, "gene": gene, "cells": int(c["Nc"]), "every": every, "peak": peak, "end": rows[-1], "trajectory": rows}
    out_dir = os.path.join(ROOT, "gpurun_out") if hip else os.path.join(ROOT, "profiles")
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, "r3u_cluster_trajectory_%s_gene%d%s.json" % (case, gene, "_hip" if hip else ""))
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(case, "gene", gene, "peak", peak, "end", rows[-1])
    for r in rows[::max(1, len(rows) // 24)]:
        print("  step %5d lr %.3f fp32 oracle: shift %.3g cells>1e-4 %d%s" % (r["step"], r["lr"], r["own_parameter_shift"], r["cells_beyond_1e-4"],
              " | HIP: shift %.3g cells>1e-4 %d" % (r["hip_own_parameter_shift"], r["hip_cells_beyond_1e-4"]) if "hip_cells_beyond_1e-4" in r else ""))


if __name__ == "__main__":
    main()
