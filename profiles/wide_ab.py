"""Wide designs at the C3 shape: MFMA tile kernel (default) vs the round-1 LDS-broadcast variants (BRIE_WIDE_PATH=lds)
vs the narrow reference model (Kc = 3, Kg = 0), INTERLEAVED in one process because the step time of one and the same
kernel drifts by up to +-10 % over seconds on these boxes (profiles/history/r02h_alloc_cycles.log: identical addresses, 8.0 .. 9.8 ms).

    python profiles/wide_ab.py [--rounds 4] [--steps 6]
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
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--cases", default="16:0,32:0,64:0,3:8,3:16,3:32,3:64,32:32")
    args = ap.parse_args()
    import torch
    from brie_amd import _capi
    Nc, Ng = 50000, 20000
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    layers = [torch.poisson(torch.full((Nc, Ng), 1.5, device=dev), generator=g) for _ in range(2)]

    def make(Kc, Kg, path):
        if path == "lds":
            os.environ["BRIE_WIDE_PATH"] = "lds"
        else:
            os.environ.pop("BRIE_WIDE_PATH", None)
            os.environ["BRIE_TILE_MIN_KG"] = "5"           # A/B: also the small gene-feature sets on the tile kernel
        sh = _capi.Shard(Nc, Ng, Kc, n_layers=2, seed=1, Kg=Kg)
        for l in range(2):
            sh.upload(_capi.COUNT1 + l, layers[l])
        sh.add_pseudo_count(0.01)
        gg = torch.Generator(device=dev)
        gg.manual_seed(7)
        if Kc:
            sh.upload(_capi.XC, torch.randn(Nc, Kc, generator=gg, device=dev))
        if Kg:
            sh.upload(_capi.XG, (torch.randn(Ng, Kg, generator=gg, device=dev) * 0.3).cpu().numpy())
        sh.init_state()
        sh.step(2, 0.005, 1, trace=False)
        sh.synchronize()
        return sh

    def timed(sh):
        t0 = time.perf_counter()
        sh.step(args.steps, 0.005, 1, trace=False)
        sh.synchronize()
        return (time.perf_counter() - t0) / args.steps * 1e3

    ref = make(3, 0, "tile")
    out = []
    for case in args.cases.split(","):
        Kc, Kg = (int(x) for x in case.split(":"))
        a, b = make(Kc, Kg, "tile"), make(Kc, Kg, "lds")
        t = {"ref": [], "tile": [], "lds": []}
        for _ in range(args.rounds):
            t["ref"].append(timed(ref)); t["tile"].append(timed(a)); t["lds"].append(timed(b))
        la, lb = a.step(1, 0.005, 1)[0], b.step(1, 0.005, 1)[0]
        r = {"Kc": Kc, "Kg": Kg, "ms_ref_Kc3": round(float(np.median(t["ref"])), 3),
             "ms_tile": round(float(np.median(t["tile"])), 3), "ms_lds": round(float(np.median(t["lds"])), 3),
             "loss_rel_diff": float(abs(la - lb) / abs(lb))}
        r["tile_over_ref"] = round(r["ms_tile"] / r["ms_ref_Kc3"], 3)
        r["lds_over_ref"] = round(r["ms_lds"] / r["ms_ref_Kc3"], 3)
        out.append(r)
        print(json.dumps(r), flush=True)
        a.close(); b.close()
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
