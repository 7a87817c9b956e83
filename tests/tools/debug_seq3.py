import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import util
from tests.test_gpu_parity import _op_sequences
from brie_amd import _capi

i, Nc, Ng, Kc, L, sparse, f32, ops = _op_sequences(40)[3]
print(i, Nc, Ng, Kc, L, sparse, f32, ops)
rng = np.random.default_rng(900 + i)
P = util.problem(Nc, Ng, Kc, L, seed=300 + i)
P["counts"] = [c.copy() for c in P["counts"]]
for _ in range(int(rng.integers(1, 6))):
    P["counts"][int(rng.integers(0, L))][int(rng.integers(0, Nc)), int(rng.integers(0, Ng))] = float(rng.integers(256, 3000))
P["counts_pc"] = util.add_pseudo_count(P["counts"], 0.01)
o = util.oracle_model(P, Nc, Ng, Kc, 40 + i, np.float32)
o64 = util.oracle_model(P, Nc, Ng, Kc, 40 + i, np.float64)
sh = _capi.Shard(Nc, Ng, Kc, n_layers=L, has_efflen=P["effLen"] is not None, seed=40 + i)
sh.set_count_storage(1)
for l in range(L):
    sh.upload(_capi.COUNT1 + l, P["counts"][l])
sh.add_pseudo_count(0.01)
sh.upload(_capi.XC, P["Xc"])
sh.init_state()


def report(tag):
    so, sd, s64 = util.oracle_state(o), util.device_state(sh), util.oracle_state(o64)
    for k in ("Z_loc", "Z_std_log", "Wc_loc", "intercept", "sigma_log"):
        d = np.abs(so[k].astype(np.float64) - sd[k]); d64 = np.abs(so[k].astype(np.float64) - s64[k])
        idx = np.unravel_index(d.argmax(), d.shape)
        print(tag, k, "max |hip - o32| %.3g at %s ; |o32 - o64| there %.3g, max %.3g" % (d.max(), idx, d64[idx], d64.max()))
    d = np.abs(so["Z_loc"].astype(np.float64) - sd["Z_loc"]); r, j = np.unravel_index(d.argmax(), d.shape)
    print("   element", (r, j), "counts", [float(P["counts"][l][r, j]) for l in range(L)], "mask", bool(o.gene_active[j]) if hasattr(o, "gene_active") and o.gene_active is not None else None,
          "Z_loc o32/hip/o64", float(so["Z_loc"][r, j]), float(sd["Z_loc"][r, j]), float(s64["Z_loc"][r, j]))


for op in ops:
    if op == "step":
        n, mc = int(rng.integers(1, 4)), int(rng.choice([1, 3, 2]))
        sh.step(n, 0.01, mc); o.minimize(P["counts_pc"], P["Xc"], n, 0.01, mc); o64.minimize(P["counts_pc"], P["Xc"], n, 0.01, mc)
        report("step n=%d mc=%d" % (n, mc))
    elif op == "mask":
        if not o.lg_hist:
            continue
        mask = rng.random(Ng) < rng.choice([0.1, 0.5, 0.9])
        if Ng > 300:
            mask[256:300] = False
        o.gene_active = mask.copy(); o64.gene_active = mask.copy()
        sh.set_gene_mask(mask)
        print("mask active", int(mask.sum()))
    elif op == "unmask":
        o.gene_active = np.ones(Ng, bool); o64.gene_active = np.ones(Ng, bool); sh.set_gene_mask(None)
    elif op == "reset":
        o.reset_optimizer(); o64.reset_optimizer(); sh.reset_optimizer()
    elif op == "tiling":
        sh.set_tiling(int(rng.choice([16, 32, 256])))
report("end")
sh.close()
