#!/usr/bin/env python
"""Result export (Psi / Z_std / Psi95CI / Z_loc -> pageable host arrays, 16 GB at configs[2]) A/B in ONE process, alone
and underneath the 500-draw loss_gene pass -- the last phase of every fit (what BRIE_RV reads, model_wrap.py:28-35).

    python profiles/egress_ab.py --modes two_streams,one_stream --out gpurun_out/egress_ab.json

two_streams (default of the library): slab k on stream k & 1, export kernel of slab k + 1 enqueued before the copies of
slab k; one_stream: BRIE_IO_ONE_STREAM=1, round 2's order.  (Call r3g measured a third variant -- staged lanes with
page-locked slabs and host memcpy threads, the mirror of the ingest -- at 0.27-0.30 s tail against 0.22 s and removed it:
profiles/history/r3g_staged_egress_ab_rejected.json.)
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
    ap.add_argument("--modes", default="one_stream,two_streams")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "egress_ab.json"))
    args = ap.parse_args()
    import torch
    import bench
    from brie_amd import _capi
    cfg = dict(bench.CONFIGS[args.config])
    Nc, Ng, Kc, L = cfg["Nc"], cfg["Ng"], cfg["Kc"], cfg["L"]
    dev = torch.device("cuda", 0)
    seed = bench.config_seed(args.config)
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
    torch.cuda.empty_cache()
    sh.init_state()
    sh.step(5, 0.005, 1, trace=False)
    sh.synchronize()
    bufs = [np.empty((Nc, Ng), np.float32) for _ in range(4)]
    for b in bufs:
        b.fill(0)                                                   # first touch, as BRIE2.fit does during the optimisation
    ref = None
    rows = []
    plan = [(m, None) for m in args.modes.split(",")] * args.reps
    for mode, thr in plan:
        if mode == "one_stream":
            os.environ["BRIE_IO_ONE_STREAM"] = "1"
        else:
            os.environ.pop("BRIE_IO_ONE_STREAM", None)
        t0 = time.perf_counter(); sh.read_results_async(*bufs); sh.read_wait(); t_alone = time.perf_counter() - t0
        t0 = time.perf_counter(); sh.read_results_async(*bufs); sh.loss_gene(500); t_lg = time.perf_counter() - t0
        sh.read_wait(); t_both = time.perf_counter() - t0
        chk = [float(b[::997, ::991].astype(np.float64).sum()) for b in bufs]
        if ref is None:
            ref = [b.copy() for b in bufs] if Nc * Ng <= 1 << 28 else chk
        same = (all(np.array_equal(a, b) for a, b in zip(ref, bufs)) if isinstance(ref[0], np.ndarray) else chk == ref)
        row = {"mode": mode, "threads": thr, "export_alone_s": t_alone, "export_GBs": 4 * Nc * Ng * 4 / t_alone / 1e9,
               "loss_gene_returned_after_s": t_lg, "both_s": t_both, "tail_behind_loss_gene_s": t_both - t_lg,
               "equals_first_run": bool(same)}
        rows.append(row)
        print(json.dumps(row), flush=True)
    with open(args.out, "w") as f:
        json.dump({"config": args.config, "bytes": 4 * Nc * Ng * 4, "runs": rows}, f, indent=1)
    sh.close()


if __name__ == "__main__":
    main()
