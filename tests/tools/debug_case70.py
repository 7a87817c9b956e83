import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import util
from tests import test_gpu_parity as T
case = [c for c in T._random_cases(250, seed=424242) if c[0] == 70][0]
print(case)
i, kind, Nc, Ng, Kc, Kg, L, MC, eff = case
P = util.problem(Nc, Ng, Kc, L, seed=1000 + i)
if not eff: P["effLen"] = None
elif P["effLen"] is None: P["effLen"] = np.random.default_rng(i).uniform(50, 400, (Ng, 6)).astype(np.float32)
mode = "cell" if kind in ("cell", "wide_cell") else "gene"
seed = 5000 + i
for path in ("tile", "lds"):
    if path == "lds": os.environ["BRIE_WIDE_PATH"] = "lds"
    else: os.environ.pop("BRIE_WIDE_PATH", None)
    o = util.oracle_model(P, Nc, Ng, Kc, seed, np.float32, Kg=Kg, mode=mode)
    o64 = util.oracle_model(P, Nc, Ng, Kc, seed, np.float64, Kg=Kg, mode=mode)
    sh = util.device_shard(P, Nc, Ng, Kc, seed, Kg=Kg, mode=mode)
    for step in range(4):
        tr_o = o.minimize(P["counts_pc"], P["Xc"], 1, 0.01, MC); o64.minimize(P["counts_pc"], P["Xc"], 1, 0.01, MC)
        tr_d = sh.step(1, 0.01, MC)
        so, sd, s64 = util.oracle_state(o), util.device_state(sh), util.oracle_state(o64)
        msg = []
        for k in ("Z_loc", "Z_std_log", "Wc_loc", "intercept", "sigma_log"):
            d = np.abs(so[k].astype(np.float64) - sd[k]); d64 = np.abs(so[k].astype(np.float64) - s64[k])
            idx = np.unravel_index(d.argmax(), d.shape)
            msg.append("%s %.2e@%s (o32-o64 max %.2e, n>1e-3: %d)" % (k, d.max(), tuple(int(x) for x in idx), d64.max(), int((d > 1e-3).sum())))
        print(path, "step", step, "loss hip/o32 %.6g %.6g |" % (tr_d[0], tr_o[0]), " ; ".join(msg), flush=True)
    sh.close()
