"""How many workgroups per CU stream best?  One handle per count storage (u8; u8/u16 per quad), the occupancy cap of the
narrow step launch (unused dynamic LDS, brie_inst.hip) toggled every ten steps: 0 -> the register-limited occupancy (3 resp.
2 per CU), 2 -> two per CU, 1 -> one per CU (the default).  (The r04j / r04k logs were taken with an experiment hook that set
the padding in bytes: pad0 / pad54000 / pad64000 = the same three settings.)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["BRIE_STEP_OCCUPANCY_CAP_DYNAMIC"] = "1"
import torch
from brie_amd import _capi
Nc, Ng = 50000, 20000
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
layers = [torch.poisson(torch.full((Nc, Ng), 1.5, device=dev), generator=g) for _ in range(2)]
for hot in (0, 1):
    if hot:
        layers[0][5, ::400] = 300.0
    sh = _capi.Shard(Nc, Ng, 3, n_layers=2, seed=1)
    for l in range(2): sh.upload(_capi.COUNT1 + l, layers[l])
    sh.add_pseudo_count(0.01)
    sh.upload(_capi.XC, torch.randn(Nc, 3, generator=g, device=dev))
    sh.init_state(); sh.step(2, 0.005, 1, trace=False); sh.synchronize()
    for rnd in range(5):
        row = {"storage": sh.count_storage}
        for cap, name in ((0, "hardware"), (2, "two_per_CU"), (1, "one_per_CU")):
            os.environ["BRIE_STEP_OCCUPANCY_CAP"] = str(cap)
            t0 = time.perf_counter(); sh.step(10, 0.005, 1, trace=False); sh.synchronize()
            row[name] = round((time.perf_counter() - t0) / 10 * 1e3, 3)
        print(json.dumps(row), flush=True)
    sh.close()
