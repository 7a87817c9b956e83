#!/usr/bin/env python
"""rows_per_chunk sweep of the step kernel in ONE process (same handle, same data, interleaved repeats).

    python profiles/rpc_sweep.py --config c2 --rpc 32,40,50,64,79,80,100,128,157,160,200,256 --out gpurun_out/rpc_c2.json

rows_per_chunk is the number of cells one workgroup streams for its 256-gene block (the per-gene partial sums are formed
per chunk, then summed in fp64).  It is a function of Nc only in the library (shard-invariant bits); this script is how
that function is chosen.  Reports kernel time per launch (HIP events), workgroups, rounds at one workgroup per CU.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2")
    ap.add_argument("--rpc", default="32,40,50,64,79,80,100,128,157,160,200,256")
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--shard-of", type=int, default=0)
    ap.add_argument("--mc", type=int, default=1)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "rpc_sweep.json"))
    args = ap.parse_args()
    import torch
    import bench
    from brie_amd import _capi
    from brie_amd.sharding import gene_shard
    cfg = dict(bench.CONFIGS[args.config])
    Nc, Ng, Kc, L = cfg["Nc"], cfg["Ng"], cfg["Kc"], cfg["L"]
    g0, g1 = (0, Ng) if not args.shard_of else gene_shard(Ng, 0, args.shard_of)
    ng = g1 - g0
    dev = torch.device("cuda", 0)
    seed = bench.config_seed(args.config)
    Xc, size, layers, eff = bench.make_inputs(torch, dev, cfg, g0, g1, seed)
    sh = _capi.Shard(Nc, ng, Kc, n_layers=L, has_efflen=L == 3, seed=seed, gene_offset=g0)
    for l in range(L):
        sh.upload(_capi.COUNT1 + l, layers[l])
    sh.add_pseudo_count(0.01)
    if L == 3:
        sh.upload(_capi.EFFLEN, eff.cpu().numpy())
    if Kc:
        sh.upload(_capi.XC, Xc)
    del layers
    sh.init_state()
    alg = sh.step_algorithmic_bytes()
    gene_blocks = -(-ng // 256)
    rpcs = [int(x) for x in args.rpc.split(",") if x]
    times = {r: [] for r in rpcs}
    for rep in range(args.reps):
        for r in rpcs:
            sh.set_tiling(r)
            sh.step(5, 0.005, args.mc, trace=False)
            sh.profile_enable(True)
            sh.step(args.steps, 0.005, args.mc, trace=False)
            ms, n = sh.profile_read()
            sh.profile_enable(False)
            times[r].append(ms / n)
    rows = []
    for r in rpcs:
        n_chunks = -(-Nc // r)
        wgs = gene_blocks * n_chunks
        best = min(times[r])
        rows.append({"rows_per_chunk": r, "n_chunks": n_chunks, "workgroups": wgs, "rounds_at_256": wgs / 256.0,
                     "kernel_ms": times[r], "best_ms": best, "frac_of_8TBs": alg / (best * 1e-3) / 8e12})
        print(json.dumps(rows[-1]), flush=True)
    with open(args.out, "w") as f:
        json.dump({"config": args.config, "genes": ng, "Nc": Nc, "mc": args.mc, "algorithmic_bytes": alg, "rows": rows},
                  f, indent=1)
    sh.close()


if __name__ == "__main__":
    main()
