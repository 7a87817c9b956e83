"""Narrow step kernel at the hardware's occupancy against 2 and 1 workgroups per CU (unused dynamic LDS at launch,
brie_inst.hip::occupancy_pads).  ONE handle per model, the cap toggled every ten steps (BRIE_STEP_OCCUPANCY_CAP_DYNAMIC):
Kc = 3 on u8 counts (166 VGPRs: 3 waves / SIMD uncapped), Kc = 1 and Kc = 0 (<= 128: 4).  (r04i log: 2 per CU vs hardware.)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["BRIE_STEP_OCCUPANCY_CAP_DYNAMIC"] = "1"


def main():
    import torch
    from brie_amd import _capi
    Nc, Ng = 50000, 20000
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(1)
    layers = [torch.poisson(torch.full((Nc, Ng), 1.5, device=dev), generator=g) for _ in range(2)]
    for Kc in (3, 1, 0):
        sh = _capi.Shard(Nc, Ng, Kc, n_layers=2, seed=1)
        for l in range(2): sh.upload(_capi.COUNT1 + l, layers[l])
        sh.add_pseudo_count(0.01)
        if Kc: sh.upload(_capi.XC, torch.randn(Nc, Kc, generator=g, device=dev))
        sh.init_state(); sh.step(2, 0.005, 1, trace=False); sh.synchronize()
        for rnd in range(5):
            row = {"Kc": Kc, "storage": sh.count_storage}
            for cap, name in (("1", "one_per_CU"), ("2", "two_per_CU"), ("0", "hardware_occupancy")):
                os.environ["BRIE_STEP_OCCUPANCY_CAP"] = cap
                t0 = time.perf_counter(); sh.step(10, 0.005, 1, trace=False); sh.synchronize()
                row[name] = round((time.perf_counter() - t0) / 10 * 1e3, 3)
            print(json.dumps(row), flush=True)
        sh.close()


if __name__ == "__main__":
    main()
