"""End-to-end timing of one reference-default fit at a BASELINE config (run on the GPU box).

    python profiles/e2e_fit.py [--config c3] [--min-iter 1000] [--mc 1]

Host numpy inputs -> BRIE2.fit (6 LR stages x int(min_iter/6) steps, 500-draw loss_gene) -> BRIE_RV
(D2H of Psi, Z_std, Psi_95CI, Z_loc), i.e. what fit_BRIE_matrix does without LRT
(/root/reference/brie/models/model_wrap.py:138-146).  Prints a breakdown incl. the PCIe legs.
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
    args = ap.parse_args()
    import torch
    import bench
    from brie_amd import _capi
    import brie_amd
    dev = torch.device("cuda", 0)
    cfg = bench.CONFIGS[args.config]
    Nc, Ng, Kc, L = cfg["Nc"], cfg["Ng"], cfg["Kc"], cfg["L"]
    gx = torch.Generator(device=dev)
    gx.manual_seed(1)
    Xc = torch.zeros(Nc, Kc, device=dev)
    if Kc:
        Xc[:, 0] = (torch.rand(Nc, generator=gx, device=dev) < 0.5).float()
        if Kc > 1:
            Xc[:, 1:] = torch.randn(Nc, Kc - 1, generator=gx, device=dev)
    size = torch.exp(0.5 * torch.randn(Nc, generator=gx, device=dev))
    host = [np.empty((Nc, Ng), np.float32) for _ in range(L)]
    eff = np.zeros((Ng, 6), np.float32) if L == 3 else None
    for c0 in range(0, Ng, bench.GEN_CHUNK):
        c1 = min(c0 + bench.GEN_CHUNK, Ng)
        cnt, e = bench.gen_chunk(torch, dev, cfg, Xc, size, c0, c1, 1)
        for l in range(L):
            host[l][:, c0:c1] = cnt[l].cpu().numpy()
        if e is not None:
            eff[c0:c1] = e.cpu().numpy()
    Xc_h = Xc.cpu().numpy()
    del Xc, size
    torch.cuda.empty_cache()

    out = {"config": cfg["desc"], "min_iter": args.min_iter, "MC_size": args.mc}
    m = brie_amd.BRIE2(Nc, Ng, Kc=Kc, effLen=eff, seed=5)
    t0 = time.perf_counter()
    sh = m._ensure_shard(host, Xc_h)          # H2D of counts + pseudo-count + init (pageable numpy memory)
    sh.synchronize()
    out["upload_init_s"] = time.perf_counter() - t0
    m._pseudo_count = 0.01
    t0 = time.perf_counter()
    n = int(args.min_iter / 6)
    for lr in brie_amd.models.engine.LEARNING_RATES:
        sh.reset_optimizer()
        losses = sh.step(n, lr, args.mc)
    out["steps"] = 6 * n
    out["steps_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    lg = sh.loss_gene(500)
    out["loss_gene_500_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    for which in (_capi.PSI, _capi.Z_STD, _capi.PSI95CI, _capi.Z_LOC):
        sh.read(which)
    out["readback_4_matrices_s"] = time.perf_counter() - t0
    out["total_s"] = out["upload_init_s"] + out["steps_s"] + out["loss_gene_500_s"] + out["readback_4_matrices_s"]
    out["it_per_s_resident"] = out["steps"] / out["steps_s"]
    out["it_per_s_pcie_inclusive"] = out["steps"] / out["total_s"]
    out["final_loss"] = float(losses[-1])
    out["loss_gene_sum"] = float(lg.sum())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
