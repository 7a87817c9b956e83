#!/usr/bin/env python
"""REJECTED EXPERIMENT (round 3; the library no longer has this path, the log is profiles/history/r3a_c1_fused_finalize_ab_rejected.log).
configs[0] (200 x 500) is launch-bound: two dependent ~6 us kernels per step.  A/B of the fused finalize (the
per-gene finalize in the tail of the step kernel, one launch per step) against the two-kernel step, alternating in ONE
process: wall time per step of brie_step(trace=False) batches, with and without per-launch profiling events.

    python profiles/c1_fused_ab.py --out gpurun_out/c1_fused_ab.json
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c1")
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "c1_fused_ab.json"))
    args = ap.parse_args()
    import torch
    import bench
    from brie_amd import _capi
    cfg = dict(bench.CONFIGS[args.config])
    Nc, Ng, Kc, L = cfg["Nc"], cfg["Ng"], cfg["Kc"], cfg["L"]
    dev = torch.device("cuda", 0)
    seed = bench.config_seed(args.config)
    Xc, size, layers, eff = bench.make_inputs(torch, dev, cfg, 0, Ng, seed)
    shards = {}
    for mode in ("0", "1"):
        os.environ["BRIE_FUSED_FINALIZE"] = mode
        sh = _capi.Shard(Nc, Ng, Kc, n_layers=L, has_efflen=L == 3, seed=seed)
        for l in range(L):
            sh.upload(_capi.COUNT1 + l, layers[l])
        sh.add_pseudo_count(0.01)
        if L == 3:
            sh.upload(_capi.EFFLEN, eff.cpu().numpy())
        if Kc:
            sh.upload(_capi.XC, Xc)
        sh.init_state()
        sh.step(50, 0.005, 1, trace=False)
        sh.synchronize()
        shards[mode] = sh
    res = {"config": args.config, "steps": args.steps, "runs": []}
    for rep in range(args.reps):
        for mode in ("0", "1"):
            sh = shards[mode]
            for prof in (False, True):
                sh.profile_enable(prof)
                sh.synchronize()
                t0 = time.perf_counter()
                sh.step(args.steps, 0.005, 1, trace=False)
                sh.synchronize()
                us = (time.perf_counter() - t0) / args.steps * 1e6
                kern = None
                if prof:
                    ms, n = sh.profile_read()
                    kern = ms / n * 1e3
                sh.profile_enable(False)
                row = {"fused": mode == "1", "profiling_events": prof, "us_per_step": us, "kernel_us_by_events": kern}
                res["runs"].append(row)
                print(json.dumps(row), flush=True)
    import numpy as np
    a, b = shards["0"].read(_capi.Z_LOC), shards["1"].read(_capi.Z_LOC)
    res["bit_identical_state"] = bool(np.array_equal(a, b))
    print("bit identical:", res["bit_identical_state"])
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
