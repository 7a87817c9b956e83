"""debug helper: cell-mode fits with very few genes, step-by-step deviation device vs fp32 / fp64 oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests import util

for (i, Nc, Ng, Kc, L, MC) in [(10, 64, 1, 8, 2, 5), (17, 63, 5, 5, 2, 3)]:
    P = util.problem(Nc, Ng, Kc, L, seed=1000 + i)
    P["effLen"] = None
    seed = 5000 + i
    o32 = util.oracle_model(P, Nc, Ng, Kc, seed, np.float32, mode="cell")
    o64 = util.oracle_model(P, Nc, Ng, Kc, seed, np.float64, mode="cell")
    sh = util.device_shard(P, Nc, Ng, Kc, seed, mode="cell")
    for step in range(4):
        a = o32.minimize(P["counts_pc"], P["Xc"], 1, 0.01, MC)
        b = o64.minimize(P["counts_pc"], P["Xc"], 1, 0.01, MC)
        c = sh.step(1, 0.01, MC)
        sd = util.device_state(sh)
        print("case", i, "step", step, "loss", a, b, c)
        for k in util.STATE_KEYS:
            x32, x64 = np.asarray(getattr(o32, k), np.float64), np.asarray(getattr(o64, k), np.float64)
            if x32.size == 0:
                continue
            d1 = np.abs(sd[k] - x64); d2 = np.abs(x32 - x64)
            print("   %-10s dev-vs-64 max %.3g at %s | o32-vs-64 max %.3g" % (k, d1.max(), np.unravel_index(d1.argmax(), d1.shape), d2.max()))
