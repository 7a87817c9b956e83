"""Reproduce soak sequence 61 (seed 31338) step by step and show WHERE the loss trace departs from the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests import util, test_gpu_parity as T
from brie_amd import _capi

case = [c for c in T._op_sequences(350, seed=31338) if c[0] == 61][0]
i, Nc, Ng, Kc, L, sparse, f32, ops = case
print(case)


def run(ops, f32=f32, drop=()):
    rng = np.random.default_rng(900 + i)
    P = util.problem(Nc, Ng, Kc, L, seed=300 + i)
    if i % 2:
        P["counts"] = [c.copy() for c in P["counts"]]
        for _ in range(int(rng.integers(1, 6))):
            P["counts"][int(rng.integers(0, L))][int(rng.integers(0, Nc)), int(rng.integers(0, Ng))] = float(rng.integers(256, 3000))
        P["counts_pc"] = util.add_pseudo_count(P["counts"], 0.01)
    o = util.oracle_model(P, Nc, Ng, Kc, 40 + i, np.float32)
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
    mask = None
    for k, op in enumerate(ops):
        if op == "step":
            n, mc = int(rng.integers(1, 4)), int(rng.choice([1, 3, 2]))
            td, to = sh.step(n, 0.01, mc), o.minimize(P["counts_pc"], P["Xc"], n, 0.01, mc)
            win = sh.read_loss_window(1)[0]
            lg = np.asarray(o.lg_hist[-1])
            bad = np.where(np.abs(win - lg) > 1e-3 * np.maximum(1, np.abs(lg)))[0]
            print(k, op, "n", n, "mc", mc, "trace hip", td, "oracle", to, "| genes whose last loss differs:", bad[:20],
                  "" if mask is None else ("frozen? %s" % (~mask[bad[:20]])), "hip", win[bad[:6]], "oracle", lg[bad[:6]])
        elif op == "mask":
            if not o.lg_hist:
                continue
            mask = rng.random(Ng) < rng.choice([0.1, 0.5, 0.9])
            if Ng > 300:
                mask[256:300] = False
            print(k, op, "active", int(mask.sum()), "of", Ng, "gene 256 active:", bool(mask[256]) if Ng > 256 else None)
            if "mask" in drop:
                continue
            o.gene_active = mask.copy()
            sh.set_gene_mask(mask)
        elif op == "loss_gene":
            a, b = sh.loss_gene(2), o.eval_loss_gene(P["counts_pc"], P["Xc"], 2)
            if "loss_gene" in drop:
                pass
            print(k, op, "max rel diff", float(np.max(np.abs(a - b) / np.maximum(1, np.abs(b)))))
        elif op == "tiling":
            r = int(rng.choice([16, 32, 256]))
            print(k, op, r)
            if "tiling" not in drop:
                sh.set_tiling(r)
        elif op == "reset":
            o.reset_optimizer(); sh.reset_optimizer()
        elif op == "unmask":
            o.gene_active = np.ones(Ng, bool); sh.set_gene_mask(None); mask = None
    sh.close()


print("=== as in the soak"); run(ops)
print("=== without the tiling change"); run(ops, drop=("tiling",))
print("=== compact storage allowed"); run(ops, f32=False)
os.environ["BRIE_PACK_ACTIVE"] = "0"
print("=== packing off"); run(ops)
