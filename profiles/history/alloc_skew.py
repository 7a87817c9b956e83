"""Step time of the headline shape in a FRESH process (one measurement per process): is the 8.4 / 10.0 ms bimodality of
the step kernel a matter of where hipMalloc puts the six state arrays?   python profiles/alloc_skew.py [--kc 3]
Environment: BRIE_STATE_SLAB=1 [BRIE_SLAB_SKEW=bytes] carves the arrays out of one allocation."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--kc", type=int, default=3)
    args = ap.parse_args()
    import torch
    from brie_amd import _capi
    Nc, Ng, Kc = 50000, 20000, args.kc
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    sh = _capi.Shard(Nc, Ng, Kc, n_layers=2, seed=1)
    for l in range(2):
        sh.upload(_capi.COUNT1 + l, torch.poisson(torch.full((Nc, Ng), 1.5, device=dev), generator=g))
    sh.add_pseudo_count(0.01)
    sh.upload(_capi.XC, torch.randn(Nc, Kc, generator=g, device=dev))
    sh.init_state()
    sh.step(3, 0.005, 1, trace=False)
    sh.synchronize()
    sh.profile_enable(True)
    t0 = time.perf_counter()
    sh.step(args.steps, 0.005, 1, trace=False)
    sh.synchronize()
    wall = (time.perf_counter() - t0) / args.steps * 1e3
    ms, n = sh.profile_read()
    print(json.dumps({"slab": os.environ.get("BRIE_STATE_SLAB", "0"), "skew": os.environ.get("BRIE_SLAB_SKEW", "0"),
                      "kernel_ms": round(ms / n, 3), "step_ms": round(wall, 3), "storage": sh.count_storage}))


if __name__ == "__main__":
    main()
