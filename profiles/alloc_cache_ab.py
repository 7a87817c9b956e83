#!/usr/bin/env python
"""Sequential fits of one size: what creating the next shard costs with and without the one-generation block cache of
brie_destroy (include/brie_amd.h: brie_trim_memory).  Alternating in ONE process at a BASELINE shape:

    python profiles/alloc_cache_ab.py --config c3 --out gpurun_out/alloc_cache_ab_c3.json
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
    ap.add_argument("--config", default="c3")
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "alloc_cache_ab.json"))
    args = ap.parse_args()
    import bench
    from brie_amd import _capi
    cfg = dict(bench.CONFIGS[args.config])
    Nc, Ng, Kc, L = cfg["Nc"], cfg["Ng"], cfg["Kc"], cfg["L"]
    rows = []

    def cycle(label, trim):
        t0 = time.perf_counter()
        sh = _capi.Shard(Nc, Ng, Kc, n_layers=L, has_efflen=L == 3, seed=1)
        sh.synchronize()
        t1 = time.perf_counter()
        sh.close()
        t2 = time.perf_counter()
        if trim:
            _capi.trim_memory()
        t3 = time.perf_counter()
        row = {"what": label, "create_s": t1 - t0, "destroy_s": t2 - t1, "trim_s": t3 - t2}
        rows.append(row)
        print(json.dumps(row), flush=True)

    cycle("first creation of the process", trim=True)
    for _ in range(args.reps):
        cycle("after a trimmed predecessor (hipFree + hipMalloc again)", trim=True)
        cycle("creation that leaves its arrays behind", trim=False)
        cycle("successor of the same size (arrays taken from the cache)", trim=True)
    with open(args.out, "w") as f:
        json.dump({"config": args.config, "shape": [Nc, Ng], "runs": rows}, f, indent=1)


if __name__ == "__main__":
    main()
