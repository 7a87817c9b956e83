"""End-to-end LRT at the headline size (run on the GPU box): fitBRIE on 50k x 20k synthetic counts with planted
cell-feature effects, ELBO-gain test of feature 0 -> detection power and false-discovery proportion against the
planted truth, plus wall time.  This is the product's actual deliverable (model_wrap.py:148-196) at BASELINE scale.

    python profiles/e2e_lrt_c3.py [--config c3] [--min-iter 1000] [--mc 1]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3")
    ap.add_argument("--min-iter", type=int, default=1000)
    ap.add_argument("--max-iter", type=int, default=5000)
    ap.add_argument("--mc", type=int, default=1)
    ap.add_argument("--common-noise", action="store_true", help="fit_BRIE_matrix(common_noise=True)")
    ap.add_argument("--no-batch-conv", action="store_true", help="one global convergence rule instead of per batch")
    ap.add_argument("--verbose", action="store_true", help="print every fit's own wall time (stderr-safe: JSON stays the last line)")
    args = ap.parse_args()
    import torch
    import bench
    import brie_amd
    from tests.fakes import FakeAnnData
    from tests.test_gpu_fullsize import _generate
    dev = torch.device("cuda", 0)
    cfg = dict(bench.CONFIGS[args.config])
    Nc, Ng, Kc = cfg["Nc"], cfg["Ng"], cfg["Kc"]
    seed = 31
    Xc, layers = _generate(torch, dev, cfg, seed)
    W_true = np.zeros((Kc, Ng), np.float32)               # the first draws of gen_chunk's per-chunk generator
    for c0 in range(0, Ng, bench.GEN_CHUNK):
        n = min(c0 + bench.GEN_CHUNK, Ng) - c0
        g = torch.Generator(device=dev)
        g.manual_seed(seed * 1000003 + c0)
        W = torch.randn(Kc, n, generator=g, device=dev) * (torch.rand(Kc, n, generator=g, device=dev) < 0.2)
        W_true[:, c0:c0 + n] = W.cpu().numpy()
    ad = FakeAnnData({'isoform1': layers[0], 'isoform2': layers[1]})
    extra = dict(conv_batch_genes=None) if args.no_batch_conv else {}
    t0 = time.perf_counter()
    res = brie_amd.fitBRIE(ad, Xc=Xc.cpu().numpy(), LRT_index=[0], layer_keys=['isoform1', 'isoform2'],
                           min_iter=args.min_iter, max_iter=args.max_iter, MC_size=args.mc, seed=7, verbose=args.verbose,
                           common_noise=args.common_noise, **extra)
    wall = time.perf_counter() - t0
    truth = W_true[0] != 0
    fdr = res.fdr[:, 0]
    called = fdr < 0.05
    est = res.cell_coeff[0]
    strong = np.abs(W_true[0]) > 0.5
    out = {
        "config": cfg["desc"], "min_iter": args.min_iter, "max_iter": args.max_iter, "MC_size": args.mc,
        "common_noise": args.common_noise, "per_batch_convergence": not args.no_batch_conv,
        "wall_s_fitBRIE_base_plus_one_test_fit": wall,
        "genes_with_planted_effect": int(truth.sum()), "called_fdr_lt_0.05": int(called.sum()),
        "true_positive_rate": float((called & truth).sum() / max(truth.sum(), 1)),
        "true_positive_rate_abs_effect_gt_0.5": float((called & strong).sum() / max(strong.sum(), 1)),
        "false_discovery_proportion": float((called & ~truth).sum() / max(called.sum(), 1)),
        "corr_estimated_vs_planted_effect_on_effect_genes": float(np.corrcoef(est[truth], W_true[0][truth])[0, 1]),
        "median_abs_error_effect_genes": float(np.median(np.abs(est[truth] - W_true[0][truth]))),
        "median_abs_estimate_null_genes": float(np.median(np.abs(est[~truth]))),
        "ELBO_gain_null_genes_quantiles_5_50_95": [float(x) for x in np.percentile(res.ELBO_gain[~truth, 0], [5, 50, 95])],
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
