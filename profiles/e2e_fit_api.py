"""End-to-end timing of ONE BRIE2.fit + BRIE_RV through the public API at a BASELINE config (run on the GPU box):
host numpy count layers -> upload, 996 staged steps, 500-draw loss_gene, Psi / Z_std / Psi95CI / Z_loc on the host
(what fit_BRIE_matrix does without LRT, /root/reference/brie/models/model_wrap.py:138-146).

    python profiles/e2e_fit_api.py [--config c3] [--no-prefetch]

--no-prefetch: the round-1 order (results read one by one after loss_gene, into fresh pageable arrays);
default: results stream out in one pass on a second stream while loss_gene runs, into arrays page-locked
during the fit (brie_read_results_async).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3")
    ap.add_argument("--min-iter", type=int, default=1000)
    ap.add_argument("--mc", type=int, default=1)
    ap.add_argument("--no-prefetch", action="store_true")
    ap.add_argument("--ab-export", type=int, default=0,
                    help="N: run N fits in this process, alternating the export order (BRIE_IO_ONE_STREAM = round 2's single "
                         "stream / unset = kernel of slab k + 1 ahead of the copies of slab k on two streams)")
    args = ap.parse_args()
    import torch
    import bench
    import brie_amd
    from tests.test_gpu_fullsize import _generate
    dev = torch.device("cuda", 0)
    cfg = bench.CONFIGS[args.config]
    Nc, Ng, Kc, L = cfg["Nc"], cfg["Ng"], cfg["Kc"], cfg["L"]
    gen = _generate(torch, dev, cfg, 1, with_eff=True)
    Xc_h = gen[0].cpu().numpy()
    host = [x.cpu().numpy() for x in gen[1]]
    eff = gen[2].cpu().numpy() if gen[2] is not None else None
    del gen
    torch.cuda.empty_cache()

    for rep in range(args.ab_export):
        mode = "one_stream" if rep % 2 == 0 else "two_streams"
        if mode == "one_stream":
            os.environ["BRIE_IO_ONE_STREAM"] = "1"
        else:
            os.environ.pop("BRIE_IO_ONE_STREAM", None)
        t0 = time.perf_counter()
        m = brie_amd.BRIE2(Nc, Ng, Kc=Kc, effLen=eff, seed=5)
        m.fit(host, Xc=Xc_h, min_iter=args.min_iter, max_iter=args.min_iter, MC_size=args.mc, pseudo_count=0.01, verbose=False)
        rv = brie_amd.BRIE_RV(m)
        tm = m.timing
        print(json.dumps({"export": mode, "total_s": round(time.perf_counter() - t0, 3), "loss_gene_s": round(tm["loss_gene_s"], 3),
                          "read_wait_s": round(tm["read_wait_s"], 3), "stage_s": [round(x, 3) for x in tm["stage_s"]],
                          "upload_s": round(tm["of_which_upload_s"], 3)}), flush=True)
        m.close()
        del rv
    if args.ab_export:
        return
    out = {"config": cfg["desc"], "min_iter": args.min_iter, "MC_size": args.mc, "prefetch_results": not args.no_prefetch}
    t0 = time.perf_counter()
    m = brie_amd.BRIE2(Nc, Ng, Kc=Kc, effLen=eff, seed=5)
    m.fit(host, Xc=Xc_h, min_iter=args.min_iter, max_iter=args.min_iter, MC_size=args.mc, pseudo_count=0.01,
          verbose=False, prefetch_results=not args.no_prefetch)
    out["fit_s"] = time.perf_counter() - t0
    out["fit_breakdown"] = dict(m.timing)
    t1 = time.perf_counter()
    rv = brie_amd.BRIE_RV(m)
    out["BRIE_RV_s"] = time.perf_counter() - t1
    out["total_s"] = time.perf_counter() - t0
    out["steps"] = 6 * int(args.min_iter / 6)
    out["it_per_s_pcie_inclusive"] = out["steps"] / out["total_s"]
    out["psi_checksum"] = float(np.asarray(rv.Psi, np.float64).sum())
    out["loss_gene_sum"] = float(np.asarray(rv.loss_gene, np.float64).sum())
    m.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
