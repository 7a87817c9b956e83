#!/usr/bin/env python
"""A few launches of ONE non-headline kernel at a BASELINE shape, to be run under rocprofv3 (kernel trace or --pmc):

    python3 profiles/counter_workload.py --what loss_gene      # loss_gene_eval, 500 draws, configs[2] shape
    python3 profiles/counter_workload.py --what tile_kc48      # elbo_adam_step_tile, 48 cell features, configs[2] shape
    python3 profiles/counter_workload.py --what tile_kg32      # elbo_adam_step_tile, Kc = 3 + 32 gene features
    python3 profiles/counter_workload.py --what c2_step        # elbo_adam_step on configs[1] (10k x 5k, effLen)
    python3 profiles/counter_workload.py --what c3_step        # the headline kernel (reference point)

Prints one JSON line with the wall time per launch measured around the launches (stream synchronised)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", required=True, choices=["loss_gene", "tile_kc48", "tile_kg32", "c2_step", "c3_step", "c5_step"])
    ap.add_argument("--launches", type=int, default=4)
    ap.add_argument("--draws", type=int, default=500)
    ap.add_argument("--genes", type=int, default=0, help="use only the first N genes (shorter counter runs)")
    ap.add_argument("--mc", type=int, default=1, help="MC_size of the step kernels (3 = the brie-quant default)")
    args = ap.parse_args()
    import torch
    import bench
    from brie_amd import _capi
    name = {"c2_step": "c2", "c5_step": "c5"}.get(args.what, "c3")
    cfg = dict(bench.CONFIGS[name])
    if args.what == "tile_kc48":
        cfg["Kc"] = 48
    Kg = 32 if args.what == "tile_kg32" else 0
    Nc, Ng, Kc, L = cfg["Nc"], cfg["Ng"], cfg["Kc"], cfg["L"]
    ng = args.genes or Ng
    dev = torch.device("cuda", 0)
    seed = bench.config_seed(name)
    Xc, size, layers, eff = bench.make_inputs(torch, dev, cfg, 0, ng, seed)
    sh = _capi.Shard(Nc, ng, Kc, n_layers=L, has_efflen=L == 3, seed=seed, Kg=Kg)
    for l in range(L):
        sh.upload(_capi.COUNT1 + l, layers[l])
    sh.add_pseudo_count(0.01)
    if L == 3:
        sh.upload(_capi.EFFLEN, eff.cpu().numpy())
    if Kc:
        sh.upload(_capi.XC, Xc)
    if Kg:
        sh.upload(_capi.XG, np.random.default_rng(3).normal(size=(ng, Kg)).astype(np.float32) * 0.3)
    del layers
    torch.cuda.empty_cache()
    sh.init_state()
    sh.step(3, 0.005, args.mc, trace=False)
    sh.synchronize()
    t0 = time.perf_counter()
    if args.what == "loss_gene":
        for _ in range(args.launches):
            sh.loss_gene(args.draws)
    else:
        sh.step(args.launches, 0.005, args.mc, trace=False)
    sh.synchronize()
    el = (time.perf_counter() - t0) / args.launches
    print(json.dumps({"what": args.what, "shape": [Nc, ng], "Kc": Kc, "Kg": Kg, "MC_size": args.mc, "launches": args.launches,
                      "draws": args.draws if args.what == "loss_gene" else None, "s_per_launch": el,
                      "algorithmic_bytes_per_step": sh.step_algorithmic_bytes(), "storage_bytes_per_step": sh.step_storage_bytes(),
                      "count_storage": sh.count_storage}))
    sh.close()


if __name__ == "__main__":
    main()
