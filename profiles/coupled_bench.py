"""Step time of the coupled variants (gene features Kg / intercept_mode='cell') at the C3 shape (run on the GPU box).

    python profiles/coupled_bench.py [--steps 10]
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
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    import torch
    import bench
    from brie_amd import _capi
    dev = torch.device("cuda", 0)
    cfg = bench.CONFIGS["c3"]
    Nc, Ng, Kc = cfg["Nc"], cfg["Ng"], cfg["Kc"]
    gx = torch.Generator(device=dev)
    gx.manual_seed(3)
    Xc = torch.randn(Nc, Kc, generator=gx, device=dev)
    size = torch.exp(0.5 * torch.randn(Nc, generator=gx, device=dev))
    layers = [torch.empty(Nc, Ng, device=dev) for _ in range(2)]
    for c0 in range(0, Ng, bench.GEN_CHUNK):
        c1 = min(c0 + bench.GEN_CHUNK, Ng)
        cnt, _ = bench.gen_chunk(torch, dev, cfg, Xc, size, c0, c1, 3)
        for l in range(2):
            layers[l][:, c0:c1] = cnt[l]
    out = []
    for mode, Kg in [(0, 0), (1, 0), (0, 2), (0, 4), (0, 8), (0, 16), (0, 32), (0, 64), (1, 16)]:
        sh = _capi.Shard(Nc, Ng, Kc, n_layers=2, seed=1, Kg=Kg, intercept_mode=mode)
        for l in range(2):
            sh.upload(_capi.COUNT1 + l, layers[l])
        sh.add_pseudo_count(0.01)
        sh.upload(_capi.XC, Xc)
        if Kg:
            sh.upload(_capi.XG, torch.randn(Ng, Kg, generator=gx, device=dev).cpu().numpy())
        sh.init_state()
        sh.step(3, 0.005, 1, trace=False)
        sh.synchronize()
        t0 = time.perf_counter()
        sh.step(args.steps, 0.005, 1, trace=False)
        sh.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        last = sh.step(1, 0.005, 1)
        assert np.isfinite(last).all()
        out.append({"intercept_mode": "cell" if mode else "gene", "Kg": Kg, "ms_per_step": round(ms, 3)})
        print(out[-1], flush=True)
        sh.close()
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
