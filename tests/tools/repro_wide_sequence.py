"""Re-run ONE sequence of tests/tools/soak_randomised.py's wide family (python tests/tools/repro_wide_sequence.py n_wide seed index)
and say where the device and the fp32 oracle part: after every step block the largest differences of Wc_loc with their genes,
next to the same oracle in fp64 (which of the two fp32 runs is the odd one?)."""
import os
import sys

os.environ.setdefault("OMP_WAIT_POLICY", "passive")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests import util
from brie_amd import _capi

n_wide, seed, index = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng0 = np.random.default_rng(seed)
case = None
for k in range(n_wide):                          # the generator of soak_randomised.wide_sequences, verbatim
    ops = [str(rng0.choice(["step", "step", "mask", "unmask", "reset", "loss_gene", "tiling", "window", "read"])) for _ in range(8)]
    c = (1000 + k, int(rng0.integers(1, 401)), int(rng0.integers(1, 1301)), int(rng0.integers(0, 13)), int(rng0.choice([2, 3])),
         bool(rng0.random() < 0.5), bool(rng0.random() < 0.5), tuple(ops))
    if c[0] == index:
        case = c
i, Nc, Ng, Kc, L, sparse, f32, ops = case
print(case)
rng = np.random.default_rng(900 + i)
P = util.problem(Nc, Ng, Kc, L, seed=300 + i)
if i % 2:
    P["counts"] = [c.copy() for c in P["counts"]]
    for _ in range(int(rng.integers(1, 6))):
        P["counts"][int(rng.integers(0, L))][int(rng.integers(0, Nc)), int(rng.integers(0, Ng))] = float(rng.integers(256, 3000))
    P["counts_pc"] = util.add_pseudo_count(P["counts"], 0.01)
o = util.oracle_model(P, Nc, Ng, Kc, 40 + i, np.float32)
o64 = util.oracle_model(P, Nc, Ng, Kc, 40 + i, np.float64)
sh = _capi.Shard(Nc, Ng, Kc, n_layers=L, has_efflen=P["effLen"] is not None, seed=40 + i)
if f32:
    sh.set_count_storage(1)
for l in range(L):
    sh.upload(_capi.COUNT1 + l, P["counts"][l])
sh.add_pseudo_count(0.01)
if P["effLen"] is not None:
    sh.upload(_capi.EFFLEN, P["effLen"])
if Kc:
    sh.upload(_capi.XC, P["Xc"])
sh.init_state()
for k, op in enumerate(ops):
    if op == "step":
        n, mc = int(rng.integers(1, 4)), int(rng.choice([1, 3, 2]))
        td, to = sh.step(n, 0.01, mc), o.minimize(P["counts_pc"], P["Xc"], n, 0.01, mc)
        o64.minimize(P["counts_pc"], P["Xc"], n, 0.01, mc)
        Wd, Wo, W64 = sh.read(_capi.WC_LOC).astype(np.float64), np.asarray(o.Wc_loc, np.float64), np.asarray(o64.Wc_loc)
        d = np.abs(Wd - Wo)
        top = np.argsort(d.ravel())[::-1][:8]
        print(k, "step n=%d mc=%d" % (n, mc), "trace hip", td, "oracle", to)
        print("   Wc_loc |hip - o32|: max %.3g, 99.9 %% %.3g; largest at (feature, gene):" % (d.max(), np.percentile(d, 99.9)))
        for t in top:
            f, g = divmod(int(t), Ng)
            print("      (%d, %d)  hip %.7f  o32 %.7f  o64 %.7f   |hip-o64| %.3g  |o32-o64| %.3g" % (
                f, g, Wd[f, g], Wo[f, g], W64[f, g], abs(Wd[f, g] - W64[f, g]), abs(Wo[f, g] - W64[f, g])))
    elif op == "reset":
        sh.reset_optimizer(); o.reset_optimizer(); o64.reset_optimizer()
        print(k, op)
    elif op == "loss_gene":
        a, b = sh.loss_gene(2), o.eval_loss_gene(P["counts_pc"], P["Xc"], 2)
        o64.eval_loss_gene(P["counts_pc"], P["Xc"], 2)
        print(k, op, "max rel diff", float(np.max(np.abs(a - b) / np.maximum(1, np.abs(b)))))
    elif op == "tiling":
        r = int(rng.choice([16, 32, 256]))
        sh.set_tiling(r)
        print(k, op, r)
    elif op in ("mask", "unmask", "window", "read"):
        print(k, op, "(consumes no state here)" if op != "mask" else "")
        if op == "mask" and o.lg_hist:
            mask = rng.random(Ng) < rng.choice([0.1, 0.5, 0.9])
            if Ng > 300:
                mask[256:300] = False
            o.gene_active = mask.copy(); o64.gene_active = mask.copy(); sh.set_gene_mask(mask)
        if op == "unmask":
            o.gene_active = np.ones(Ng, bool); o64.gene_active = np.ones(Ng, bool); sh.set_gene_mask(None)
