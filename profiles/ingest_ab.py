#!/usr/bin/env python
"""Host -> device ingest of the count layers, A/B in ONE process: the staged pipeline (host threads convert row slabs to
u16 in page-locked buffers, asynchronous copies, a kernel writes the tiled layer) against the plain strided copies out
of pageable memory (BRIE_INGEST=direct), alternating, on configs[2]-sized layers (2 x 50k x 20k fp32 = 8 GB).

    python profiles/ingest_ab.py --out gpurun_out/ingest_ab.json [--config c3] [--reps 3] [--threads 2,4,6,8]

Reference being replaced: the densify + cast of model_wrap.py:108-111 followed by TensorFlow's own host -> device copy.
"""
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
    ap.add_argument("--config", default="c3")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--threads", default="")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "ingest_ab.json"))
    args = ap.parse_args()
    import torch
    import bench
    from brie_amd import _capi
    cfg = dict(bench.CONFIGS[args.config])
    Nc, Ng, Kc, L = cfg["Nc"], cfg["Ng"], cfg["Kc"], cfg["L"]
    dev = torch.device("cuda", 0)
    seed = bench.config_seed(args.config)
    Xc, size, layers, eff = bench.make_inputs(torch, dev, cfg, 0, Ng, seed)
    host = [x.cpu().numpy() for x in layers]                  # pageable, as the API hands them over
    del layers
    torch.cuda.empty_cache()
    res = {"config": args.config, "shape": [Nc, Ng], "layers": L, "bytes": int(sum(h.nbytes for h in host)),
           "host_cores": os.cpu_count(), "runs": []}

    def one(mode, threads=None):
        os.environ["BRIE_INGEST"] = mode
        if threads:
            os.environ["BRIE_INGEST_THREADS"] = str(threads)
        else:
            os.environ.pop("BRIE_INGEST_THREADS", None)
        t0 = time.perf_counter()
        sh = _capi.Shard(Nc, Ng, Kc, n_layers=L, has_efflen=L == 3, seed=seed)
        t1 = time.perf_counter()
        for l in range(L):
            sh.upload(_capi.COUNT1 + l, host[l])
        sh.synchronize()
        t2 = time.perf_counter()
        sh.add_pseudo_count(0.01)
        sh.synchronize()
        t3 = time.perf_counter()
        out = {"mode": mode, "threads": threads, "create_s": t1 - t0, "upload_s": t2 - t1, "tiers_s": t3 - t2,
               "GBs": res["bytes"] / (t2 - t1) / 1e9, "storage": sh.count_storage}
        return sh, out

    # correctness once: the staged layer equals the plain copy bit for bit (first layer, read back)
    sh, _ = one("direct")
    ref = sh.read(_capi.COUNT1)
    sh.close()
    sh, _ = one("staged")
    got = sh.read(_capi.COUNT1)
    sh.close()
    res["staged_equals_direct"] = bool(np.array_equal(ref, got))
    del ref, got
    plan = []
    for _ in range(args.reps):
        plan += [("direct", None), ("staged", None)]
    for t in [int(x) for x in args.threads.split(",") if x]:
        plan += [("staged", t)]
    for mode, thr in plan:
        sh, out = one(mode, thr)
        sh.close()
        res["runs"].append(out)
        print(json.dumps(out), flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
