#!/usr/bin/env python
"""The PSI parity statistics of one case as a TIME SERIES over the last learning-rate stage, not as the end-of-fit
snapshot the rule uses (under MC noise the fp32-vs-fp64 difference of a gene comes and goes).

    python profiles/parity_over_time.py --oracles mid_cli_96_s3      (CPU, build container: fp32 + fp64 C restatement;
                                                                      fp64 checkpoints -> profiles/_psi_cache)
    python profiles/parity_over_time.py --hip mid_cli_96_s3          (GPU box: the HIP path against those checkpoints)
At every checkpoint (every 49 steps of the sixth stage and its last step): entries beyond 1e-4, displaced genes (own
parameter off by > 4e-4), clustered genes -- for the fp32 oracle and for the HIP path, each against the fp64 run.
The oracle is the checker here, never the thing measured."""
import argparse
import json
import os
import sys

import numpy as np

os.environ.setdefault("OMP_WAIT_POLICY", "passive")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from profiles import psi_delta as pd       # noqa: E402
from tests import util                     # noqa: E402

EVERY = 49


def plan(case):
    """[(stage index, steps to run, lr, checkpoint after?)] -- checkpoints only in the last stage."""
    c = pd.CASES[case]
    out = []
    sched = pd.schedule(c["min_iter"])
    for i, (n, lr) in enumerate(sched):
        if i < len(sched) - 1:
            out.append((i, n, lr, False))
        else:
            done = 0
            while done < n:
                k = min(EVERY, n - done)
                out.append((i, k, lr, True))
                done += k
    return out


def stats(psi, par, psi64, par64, Nc):
    d = np.abs(np.asarray(psi, np.float64) - np.asarray(psi64, np.float64))
    ex = d > 1e-4
    sh = util.gene_shift(par, par64)
    disp = sh > util.GENE_SHIFT
    clus = ~disp & (ex.sum(0) > max(5, int(1e-3 * Nc)))
    return {"entries_gt_1e-4": int(ex.sum()), "displaced_genes": int(disp.sum()), "clustered_genes": int(clus.sum()),
            "entries_in_quiet_genes": int(ex[:, ~(disp | clus)].sum()), "largest_shift": float(sh.max())}


def cache_path(case):
    return os.path.join(pd.CACHE, "%s_overtime.npz" % case)


def run_oracles(case):
    from oracle.c_oracle import COracle
    P, c = pd.problem(case)
    runs = {dt: COracle(P["counts_pc"], P["Xc"], effLen=P["effLen"], seed=pd.model_seed(case), dtype=dt) for dt in (np.float32, np.float64)}
    for o in runs.values():
        o.set_threads(8)
    step, last, saved, series = 0, -1, {}, []
    for stage, n, lr, ck in plan(case):
        if stage != last:
            for o in runs.values():
                o.reset_optimizer()
            last = stage
        for o in runs.values():
            o.minimize(n, lr, c["MC"])
        step += n
        if ck:
            a, b = runs[np.float32], runs[np.float64]
            pb = util.run_params(b)
            series.append(dict(step=step, **stats(a.Psi, util.run_params(a), b.Psi, pb, c["Nc"])))
            k = len(series) - 1
            saved["psi_%d" % k] = np.asarray(b.Psi, np.float32)
            for name, v in pb.items():
                saved["%s_%d" % (name, k)] = np.array(v, copy=True)      # (run_params may hand out the oracle's live buffers)
            print(series[-1], flush=True)
    os.makedirs(pd.CACHE, exist_ok=True)
    np.savez(cache_path(case), steps=np.array([s["step"] for s in series]), o32_series=json.dumps(series), **saved)


def run_hip(case, out):
    from brie_amd import _capi
    z = np.load(cache_path(case))
    o32 = json.loads(str(z["o32_series"]))
    P, c = pd.problem(case)
    sh = util.device_shard(P, c["Nc"], c["Ng"], c["Kc"], pd.model_seed(case))
    step, last, k, series = 0, -1, 0, []
    for stage, n, lr, ck in plan(case):
        if stage != last:
            sh.reset_optimizer()
            last = stage
        sh.step(n, lr, c["MC"], trace=False)
        step += n
        if ck:
            assert int(z["steps"][k]) == step
            par64 = {name: z["%s_%d" % (name, k)] for name in ("Wc_loc", "intercept", "sigma_log")}
            series.append(dict(step=step, **stats(sh.read(_capi.PSI), util.run_params(sh), z["psi_%d" % k], par64, c["Nc"])))
            k += 1
    sh.close()
    keys = ("entries_gt_1e-4", "displaced_genes", "clustered_genes", "entries_in_quiet_genes")
    res = {"case": case, "desc": c["desc"], "checkpoints": len(series), "every": EVERY,
           "what": "statistics of the PSI parity rule at every checkpoint of the LAST learning-rate stage, each fp32 run against the fp64 oracle",
           "mean_over_checkpoints": {"hip": {q: float(np.mean([s[q] for s in series])) for q in keys},
                                     "fp32_oracle": {q: float(np.mean([s[q] for s in o32])) for q in keys}},
           "end_of_fit": {"hip": series[-1], "fp32_oracle": o32[-1]},
           "series": {"hip": series, "fp32_oracle": o32}}
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({"mean_over_checkpoints": res["mean_over_checkpoints"], "end_of_fit": res["end_of_fit"]}, indent=1))
    for h, o in zip(series, o32):
        print("step %5d  HIP: entries %6d displaced %3d clustered %2d quiet-gene entries %4d | fp32 oracle: %6d %3d %2d %4d" % (
            h["step"], h["entries_gt_1e-4"], h["displaced_genes"], h["clustered_genes"], h["entries_in_quiet_genes"],
            o["entries_gt_1e-4"], o["displaced_genes"], o["clustered_genes"], o["entries_in_quiet_genes"]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--oracles", default=None)
    ap.add_argument("--hip", default=None)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    if args.oracles:
        run_oracles(args.oracles)
    if args.hip:
        run_hip(args.hip, args.out or os.path.join(ROOT, "gpurun_out", "parity_over_time_%s.json" % args.hip))


if __name__ == "__main__":
    main()
