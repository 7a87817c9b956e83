#!/usr/bin/env python
"""loss_gene_eval (the 500-draw forward pass that ends every fit, model_TFProb.py:261-264) A/B between two builds of
the library, alternating processes on one box:

    python profiles/loss_gene_ab.py --libs default,brie_amd/lib/variants/libbrie_amd_lg_round2.so --out gpurun_out/lg_ab.json

Each process: configs[2] shape, a few steps, then `reps` x brie_loss_gene(500) timed; prints per-gene checksums so that
the two builds' VALUES can be compared too (they differ in the last bits: other rounding of log1p for |z| > 6.9)."""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(config, reps, draws):
    import torch
    import bench
    from brie_amd import _capi
    cfg = dict(bench.CONFIGS[config])
    Nc, Ng, Kc, L = cfg["Nc"], cfg["Ng"], cfg["Kc"], cfg["L"]
    dev = torch.device("cuda", 0)
    seed = bench.config_seed(config)
    Xc, size, layers, eff = bench.make_inputs(torch, dev, cfg, 0, Ng, seed)
    sh = _capi.Shard(Nc, Ng, Kc, n_layers=L, has_efflen=L == 3, seed=seed)
    for l in range(L):
        sh.upload(_capi.COUNT1 + l, layers[l])
    sh.add_pseudo_count(0.01)
    if L == 3:
        sh.upload(_capi.EFFLEN, eff.cpu().numpy())
    if Kc:
        sh.upload(_capi.XC, Xc)
    del layers
    sh.init_state()
    sh.step(20, 0.01, 1, trace=False)
    sh.synchronize()
    times, lg = [], None
    for _ in range(reps):
        sh.draw = 1000
        t0 = time.perf_counter()
        lg = sh.loss_gene(draws)
        times.append(time.perf_counter() - t0)
    print(json.dumps({"lib": os.environ.get("BRIE_AMD_LIB", "default"), "config": config, "draws": draws, "seconds": times,
                      "loss_gene_sum": float(lg.astype(np.float64).sum()), "loss_gene_first": [float(x) for x in lg[:4]]}))
    sh.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", default="default")
    ap.add_argument("--configs", default="c3,c2")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--draws", type=int, default=500)
    ap.add_argument("--worker", default=None)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "loss_gene_ab.json"))
    args = ap.parse_args()
    if args.worker:
        worker(args.worker, args.reps, args.draws)
        return
    rows = []
    for config in args.configs.split(","):
        for _ in range(args.rounds):
            for lib in args.libs.split(","):
                env = dict(os.environ)
                if lib != "default":
                    env["BRIE_AMD_LIB"] = os.path.join(ROOT, lib) if not os.path.isabs(lib) else lib
                p = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", config, "--reps", str(args.reps),
                                    "--draws", str(args.draws)], env=env, capture_output=True, text=True)
                line = [l for l in p.stdout.splitlines() if l.startswith("{")]
                if not line:
                    print(p.stderr[-1500:])
                    continue
                rows.append(json.loads(line[-1]))
                print(line[-1], flush=True)
    with open(args.out, "w") as f:
        json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
