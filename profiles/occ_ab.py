"""Narrow step kernel at 3 vs 2 workgroups per CU (the u8 instantiation compiles to 166 VGPRs = 3 waves / SIMD, the
u16 / tiered ones to 2).  Two handles on the same data, created with BRIE_STEP_OCCUPANCY_CAP=0 (occupancy left to the
hardware: 3) and unset (54 KB of unused dynamic LDS at launch: 2), stepped alternately.  (profiles/r03m_occ_ab.log was
taken with an experiment build that toggled the padding per launch on ONE handle.)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from brie_amd import _capi
    Nc, Ng = 50000, 20000
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(1)
    layers = [torch.poisson(torch.full((Nc, Ng), 1.5, device=dev), generator=g) for _ in range(2)]
    Xc = torch.randn(Nc, 3, generator=g, device=dev)

    def make():
        sh = _capi.Shard(Nc, Ng, 3, n_layers=2, seed=1)
        for l in range(2): sh.upload(_capi.COUNT1 + l, layers[l])
        sh.add_pseudo_count(0.01)
        sh.upload(_capi.XC, Xc)
        sh.init_state(); sh.step(2, 0.005, 1, trace=False); sh.synchronize()
        return sh
    # the padding of an instantiation is decided at its first launch in the process: run the uncapped handle in a child
    if len(sys.argv) > 1 and sys.argv[1] == "uncapped":
        os.environ["BRIE_STEP_OCCUPANCY_CAP"] = "0"
    sh = make()
    print(sh.count_storage, os.environ.get("BRIE_STEP_OCCUPANCY_CAP", "capped"))
    for rnd in range(6):
        t0 = time.perf_counter(); sh.step(10, 0.005, 1, trace=False); sh.synchronize()
        print(json.dumps({"ms_per_step": round((time.perf_counter() - t0) / 10 * 1e3, 3)}), flush=True)


if __name__ == "__main__":
    main()
