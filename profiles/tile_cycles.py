"""Per-block step times of the tile kernel for fresh handles (is its run-to-run spread per handle or over time?)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from brie_amd import _capi
    Nc, Ng = 50000, 20000
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(1)
    layers = [torch.poisson(torch.full((Nc, Ng), 1.5, device=dev), generator=g) for _ in range(2)]

    def make(Kc, Kg, nw):
        if nw: os.environ["BRIE_TILE_HALVES"] = str(nw)
        else: os.environ.pop("BRIE_TILE_HALVES", None)
        sh = _capi.Shard(Nc, Ng, Kc, n_layers=2, seed=1, Kg=Kg)
        for l in range(2): sh.upload(_capi.COUNT1 + l, layers[l])
        sh.add_pseudo_count(0.01)
        if Kc: sh.upload(_capi.XC, torch.randn(Nc, Kc, generator=g, device=dev))
        if Kg: sh.upload(_capi.XG, (torch.randn(Ng, Kg, generator=g, device=dev) * 0.3).cpu().numpy())
        sh.init_state(); sh.step(2, 0.005, 1, trace=False); sh.synchronize()
        return sh

    def blocks(sh, n=5, steps=6):
        out = []
        for _ in range(n):
            t0 = time.perf_counter(); sh.step(steps, 0.005, 1, trace=False); sh.synchronize()
            out.append(round((time.perf_counter() - t0) / steps * 1e3, 2))
        return out
    ref = make(3, 0, 0)
    for cyc in range(3):
        for (Kc, Kg) in ((32, 0), (3, 32)):
            for nw in (1, 2):
                sh = make(Kc, Kg, nw)
                print(json.dumps({"cycle": cyc, "Kc": Kc, "Kg": Kg, "nw": nw, "tile": blocks(sh), "ref": blocks(ref, 2)}), flush=True)
                sh.close()


if __name__ == "__main__":
    main()
