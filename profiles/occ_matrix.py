"""Which occupancy cap for which instantiation?  Step time at one / two workgroups per CU / hardware occupancy, toggled on
one handle (BRIE_STEP_OCCUPANCY_CAP_DYNAMIC), for Kc x likelihood x MC_size at 50k cells x 10k genes."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["BRIE_STEP_OCCUPANCY_CAP_DYNAMIC"] = "1"
import numpy as np
import torch
LS_ENV = [int(k) for k in os.environ.get("OCC_L", "2,3").split(",")]
from brie_amd import _capi
Nc, Ng = int(os.environ.get("OCC_NC", 50000)), int(os.environ.get("OCC_NG", 10000))
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
layers = [torch.poisson(torch.full((Nc, Ng), 1.5, device=dev), generator=g) for _ in range(max(LS_ENV))]
layers[0][5, ::400] = 300.0          # tiered counts, as in the bench
eff = np.abs(np.random.default_rng(0).normal(200, 30, size=(Ng, 6))).astype(np.float32) + 50
KCS = [int(k) for k in os.environ.get("OCC_KC", "0,1,3,5,8").split(",")]
LS = [int(k) for k in os.environ.get("OCC_L", "2,3").split(",")]
for Kc in KCS:
    for L in LS:
        for mc in (1, 3):
            sh = _capi.Shard(Nc, Ng, Kc, n_layers=L, has_efflen=(L == 3), seed=1)
            for l in range(L): sh.upload(_capi.COUNT1 + l, layers[l])
            sh.add_pseudo_count(0.01)
            if L == 3: sh.upload(_capi.EFFLEN, eff)
            if Kc: sh.upload(_capi.XC, torch.randn(Nc, Kc, generator=g, device=dev))
            sh.init_state(); sh.step(2, 0.005, mc, trace=False); sh.synchronize()
            best = {}
            for rnd in range(3):
                for cap, name in (("1", "one"), ("2", "two"), ("0", "hw")):
                    os.environ["BRIE_STEP_OCCUPANCY_CAP"] = cap
                    t0 = time.perf_counter(); sh.step(8, 0.005, mc, trace=False); sh.synchronize()
                    t = (time.perf_counter() - t0) / 8 * 1e3
                    best[name] = min(best.get(name, 1e9), t)
            print(json.dumps({"Kc": Kc, "effLen": L == 3, "MC": mc, **{k: round(v, 3) for k, v in best.items()},
                              "one_over_two": round(best["one"] / best["two"], 4)}), flush=True)
            sh.close()
