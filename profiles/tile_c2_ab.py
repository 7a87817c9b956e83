"""One measurement per process: median step time of the tile kernel for (Kc, Kg) given on the command line, next to the
narrow model, both in this process (BRIE_AMD_LIB selects the build)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from brie_amd import _capi
    Kc, Kg = int(sys.argv[1]), int(sys.argv[2])
    Nc, Ng = 50000, 20000
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(1)
    layers = [torch.poisson(torch.full((Nc, Ng), 1.5, device=dev), generator=g) for _ in range(2)]

    def make(Kc, Kg):
        sh = _capi.Shard(Nc, Ng, Kc, n_layers=2, seed=1, Kg=Kg)
        for l in range(2): sh.upload(_capi.COUNT1 + l, layers[l])
        sh.add_pseudo_count(0.01)
        if Kc: sh.upload(_capi.XC, torch.randn(Nc, Kc, generator=g, device=dev))
        if Kg: sh.upload(_capi.XG, (torch.randn(Ng, Kg, generator=g, device=dev) * 0.3).cpu().numpy())
        sh.init_state(); sh.step(2, 0.005, 1, trace=False); sh.synchronize()
        return sh

    def blk(sh, steps=6):
        t0 = time.perf_counter(); sh.step(steps, 0.005, 1, trace=False); sh.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3
    ref, a = make(3, 0), make(Kc, Kg)
    tr, ta = [], []
    for _ in range(6):
        tr.append(blk(ref)); ta.append(blk(a))
    print(json.dumps({"lib": os.path.basename(os.environ.get("BRIE_AMD_LIB", "default")), "Kc": Kc, "Kg": Kg,
                      "ref": round(float(np.median(tr)), 3), "tile": round(float(np.median(ta)), 3),
                      "ratio": round(float(np.median(ta) / np.median(tr)), 3)}))


if __name__ == "__main__":
    main()
