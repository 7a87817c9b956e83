"""C3-size fit through BRIE2.fit with the fitBRIE default: per-batch convergence (10-gene batches at Nc=50k).
Reports the distribution of per-batch n_iter and the wall time of the extension phase (run on the GPU box)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import brie_amd
from tests.test_gpu_fullsize import _generate

dev = torch.device("cuda", 0)
cfg = dict(bench.CONFIGS[os.environ.get("CONFIG", "c3")])
Nc, Ng, Kc = cfg["Nc"], cfg["Ng"], cfg["Kc"]
Xc, layers = _generate(torch, dev, cfg, 5)
m = brie_amd.BRIE2(Nc, Ng, Kc=Kc, seed=5)
t0 = time.perf_counter()
losses = m.fit(layers, Xc=Xc.cpu().numpy(), min_iter=int(os.environ.get("MIN_ITER", "1000")),
               max_iter=int(os.environ.get("MAX_ITER", "5000")), add_iter=500, epsilon_conv=1e-2, n_loss_gene=500,
               pseudo_count=0.01, verbose=False, conv_batch_genes=int(np.ceil(500000 / Nc)))
dt = time.perf_counter() - t0
vals, cnt = np.unique(m.n_iter_batch, return_counts=True)
print(json.dumps({"config": cfg["desc"], "fit_s": dt, "trace_len": int(len(losses)), "steps_run": int(996 + len(losses) - 166),
                  "n_iter_batch_hist": {int(v): int(c) for v, c in zip(vals, cnt)},
                  "mean_n_iter": float(m.n_iter_batch.mean()), "timing": {k: v for k, v in m.timing.items()},
                  "rounds": m.round_log}))
