"""The streamed export of Psi / Z_std / Psi95CI / Z_loc (16 GB at C3) on its own, next to loss_gene on its own and both
together: which one bounds the 0.8 s of the fit's last phase?"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from brie_amd import _capi
Nc, Ng = 50000, 20000
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
layers = [torch.poisson(torch.full((Nc, Ng), 1.5, device=dev), generator=g) for _ in range(2)]
sh = _capi.Shard(Nc, Ng, 3, n_layers=2, seed=1)
for l in range(2): sh.upload(_capi.COUNT1 + l, layers[l])
sh.add_pseudo_count(0.01)
sh.upload(_capi.XC, torch.randn(Nc, 3, generator=g, device=dev))
sh.init_state(); sh.step(2, 0.005, 1, trace=False); sh.synchronize()
bufs = [np.empty((Nc, Ng), np.float32) for _ in range(4)]
for b in bufs: b.fill(0)
for rnd in range(3):
    t0 = time.perf_counter(); sh.loss_gene(500); t_lg = time.perf_counter() - t0
    t0 = time.perf_counter(); sh.read_results_async(*bufs); sh.read_wait(); t_ex = time.perf_counter() - t0
    t0 = time.perf_counter(); sh.read_results_async(*bufs); sh.loss_gene(500); t_mid = time.perf_counter() - t0; sh.read_wait(); t_both = time.perf_counter() - t0
    print(json.dumps({"loss_gene_alone_s": round(t_lg, 3), "export_alone_s": round(t_ex, 3), "export_GBs": round(16 / t_ex, 1),
                      "both_s": round(t_both, 3), "loss_gene_returned_after_s": round(t_mid, 3)}), flush=True)
