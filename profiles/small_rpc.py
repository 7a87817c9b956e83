#!/usr/bin/env python
"""Small problems (the size class of real brie-quant inputs: hundreds to a few thousand cells): WALL time per Adam step --
both launches of a step, host enqueue included -- over rows_per_chunk, same handle, interleaved repeats.

    python profiles/small_rpc.py --out gpurun_out/small_rpc.json

The library's rows_per_chunk is a function of Nc only (shard-invariant bits); this script is how its small-Nc branch is
chosen.  Below ~6000 cells a launch has fewer workgroups than the chip has CUs, every wave walks rows_per_chunk / 4 rows
one after the other (~2 us each) and that serial walk, not HBM, is the step."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = {
    # name: Nc, Ng, Kc, L
    "c1_200x500": (200, 500, 0, 2),
    "300x2000_eff": (300, 2000, 1, 3),
    "1000x3000_eff": (1000, 3000, 1, 3),
    "3000x5000_eff": (3000, 5000, 2, 3),
    "5000x5000": (5000, 5000, 1, 2),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rpc", default="0,4,8,16,32,64")
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--mc", default="1,3")
    ap.add_argument("--shapes", default=",".join(SHAPES))
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "small_rpc.json"))
    args = ap.parse_args()
    import torch
    import bench
    from brie_amd import _capi
    dev = torch.device("cuda", 0)
    rpcs = [int(x) for x in args.rpc.split(",") if x != ""]
    out = {"steps": args.steps, "reps": args.reps, "unit": "us per step (wall, best of reps)", "cases": {}}
    for name in [s for s in args.shapes.split(",") if s]:
        Nc, Ng, Kc, L = SHAPES[name]
        cfg = dict(Nc=Nc, Ng=Ng, Kc=Kc, L=L, theta=1.5, depth=2.0)
        Xc, size, layers, eff = bench.make_inputs(torch, dev, cfg, 0, Ng, 777)
        sh = _capi.Shard(Nc, Ng, Kc, n_layers=L, has_efflen=L == 3, seed=5)
        for l in range(L):
            sh.upload(_capi.COUNT1 + l, layers[l])
        sh.add_pseudo_count(0.01)
        if L == 3:
            sh.upload(_capi.EFFLEN, eff.cpu().numpy())
        if Kc:
            sh.upload(_capi.XC, Xc)
        sh.init_state()
        for mc in [int(x) for x in args.mc.split(",")]:
            times = {r: [] for r in rpcs}
            for rep in range(args.reps):
                for r in rpcs:
                    if r and r > Nc:
                        continue
                    sh.set_tiling(r)
                    sh.step(50, 0.005, mc, trace=False)
                    sh.synchronize()
                    t0 = time.perf_counter()
                    sh.step(args.steps, 0.005, mc, trace=False)
                    sh.synchronize()
                    times[r].append((time.perf_counter() - t0) / args.steps * 1e6)
            row = {str(r): round(min(v), 2) for r, v in times.items() if v}
            out["cases"]["%s_mc%d" % (name, mc)] = row
            print(name, "mc", mc, row, flush=True)
        sh.close()
        del layers
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
