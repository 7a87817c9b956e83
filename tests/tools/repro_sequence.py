"""Re-run ONE sequence of the committed-family generator with a soak's seed (python tests/tools/repro_sequence.py n seed index;
soak_randomised.py draws its sequences from _op_sequences(n, seed + 1)) and say, after every step block, which Wc_loc entries
sit beyond 1e-3 of the fp32 oracle: their genes, those genes' total counts, and the fp32 oracle against the same oracle in fp64."""
import os
import sys

os.environ.setdefault("OMP_WAIT_POLICY", "passive")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests import util, test_gpu_parity as T
from brie_amd import _capi

n, seed, index = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
case = [c for c in T._op_sequences(n, seed=seed) if c[0] == index][0]
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
tot = sum(np.asarray(c).sum(axis=0) for c in P["counts"])
for k, op in enumerate(ops):
    if op == "step":
        nst, mc = int(rng.integers(1, 4)), int(rng.choice([1, 3, 2]))
        sh.step(nst, 0.01, mc); o.minimize(P["counts_pc"], P["Xc"], nst, 0.01, mc); o64.minimize(P["counts_pc"], P["Xc"], nst, 0.01, mc)
        wd, wo, w64 = sh.read(_capi.WC_LOC), np.asarray(o.Wc_loc), np.asarray(o64.Wc_loc)
        bad = np.argwhere(np.abs(wd - wo) > 1e-3)
        print("op %d: %d step(s) MC %d -> %d entries of Wc_loc beyond 1e-3 (HIP vs o32), %d (o32 vs o64), %d (HIP vs o64)"
              % (k, nst, mc, len(bad), int((np.abs(wo - w64) > 1e-3).sum()), int((np.abs(wd - w64) > 1e-3).sum())))
        for f, g in bad[:12]:
            print("   feature %2d gene %4d: HIP %+.6f o32 %+.6f o64 %+.6f; the gene's counts over all cells and layers: %g"
                  % (f, g, wd[f, g], wo[f, g], w64[f, g], tot[g]))
    elif op == "mask":
        if not o.lg_hist:
            continue
        mask = rng.random(Ng) < rng.choice([0.1, 0.5, 0.9])
        if Ng > 300:
            mask[256:300] = False
        o.gene_active = mask.copy(); o64.gene_active = mask.copy(); sh.set_gene_mask(mask)
    elif op == "unmask":
        o.gene_active = np.ones(Ng, bool); o64.gene_active = np.ones(Ng, bool); sh.set_gene_mask(None)
    elif op == "reset":
        o.reset_optimizer(); o64.reset_optimizer(); sh.reset_optimizer()
    elif op == "tiling":
        sh.set_tiling(int(rng.choice([16, 32, 256])))
    elif op == "loss_gene":
        sh.loss_gene(2); o.eval_loss_gene(P["counts_pc"], P["Xc"], 2); o64.eval_loss_gene(P["counts_pc"], P["Xc"], 2)
sh.close()
