"""Debug helper: C3-size fit for a few steps, compare scattered gene quads with the oracle."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from brie_amd import _capi
from oracle.brie_oracle import OracleBRIE2, add_pseudo_count
from tests.test_gpu_fullsize import _generate

dev = torch.device("cuda", 0)
cfg = bench.CONFIGS["c3"]
Nc, Ng, Kc = cfg["Nc"], cfg["Ng"], cfg["Kc"]
seed = 424242
STEPS = int(os.environ.get("STEPS", "12"))
Xc, layers = _generate(torch, dev, cfg, seed)
sh = _capi.Shard(Nc, Ng, Kc, n_layers=2, seed=seed)
for l in range(2):
    sh.upload(_capi.COUNT1 + l, layers[l])
sh.add_pseudo_count(0.01)
sh.upload(_capi.XC, Xc)
sh.init_state()
trace = sh.step(STEPS, 0.01, 1)
zloc = sh.read(_capi.Z_LOC); zsl = sh.read(_capi.Z_STD_LOG)
W, b, lam = sh.read(_capi.WC_LOC), sh.read(_capi.INTERCEPT), sh.read(_capi.SIGMA_LOG)
Xc_h = Xc.cpu().numpy()
print("lib", os.environ.get("BRIE_AMD_LIB", "default"), "trace", trace[:3], trace[-1])
for g0 in (0, 252, 256, 1000, 5000, 7164, 7168, 7316, 7420, 7424, 12000, 19964, 19968, 19996):
    cols = slice(g0, g0 + 4)
    cnt = add_pseudo_count([layers[l][:, cols].cpu().numpy() for l in range(2)])
    o = OracleBRIE2(Nc, 4, Kc, seed=seed, gene_offset=g0, dtype=np.float32)
    o.minimize(cnt, Xc_h, STEPS, 0.01, 1)
    out = []
    for name, d_, r_ in (("Z", zloc[:, cols], o.Z_loc), ("rho", zsl[:, cols], o.Z_std_log), ("W", W[:, cols], o.Wc_loc),
                         ("b", b[:, cols], o.intercept), ("lam", lam[:, cols], o.sigma_log)):
        out.append("%s %.2e" % (name, np.abs(d_ - r_).max()))
    print("g0 %5d block %2d lane %2d: %s" % (g0, g0 // 256, (g0 % 256) // 4, "  ".join(out)))
