"""Seeded synthetic counts for parity tests and the bench.  TEST INFRASTRUCTURE ONLY.

Recipe = SURVEY.md 8(d): generative model of
/root/reference/brie/models/simulator.py:22-69 (Psi = expit(clip(Z, +-9)),
phi ~ [Psi, 1-Psi, 1] * effLen[:, [0,4,5]], multinomial counts) with the PSI
prior of /root/reference/simulator/simuPSI.py:19-20,129-130
(logit Psi ~ N(0, theta^2), theta = 3) and the exon-skipping effective lengths
of /root/reference/brie/utils/count.py:83-95.
"""
import numpy as np
from scipy.special import expit


def se_efflen(l1, l2, l3, rlen=76, edge_hang=10, junc_hang=2):
    """count.py:88-95 -> (Ng, 6) = [iso1(g1,g2,g3), iso2(g1,g2,g3)] (io_utils.py:47-48)."""
    n = len(l1)
    eff = np.zeros((n, 6), np.float32)
    eff[:, 0] = l2 + rlen - 2 * junc_hang
    eff[:, 4] = rlen - 2 * junc_hang
    eff[:, 2] = l1 + l3 - 2 * edge_hang + 2 * junc_hang
    eff[:, 5] = eff[:, 2]
    return eff


def make_problem(Nc, Ng, Kc=0, L=2, seed=20240617, theta=1.5, depth=2.0, effect_frac=0.2):
    """Returns dict(counts=[L x (Nc,Ng) f32], Xc (Nc,Kc) f32, effLen (Ng,6)|None, truth...)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    Xc = np.zeros((Nc, Kc), np.float32)
    if Kc > 0:
        Xc[:, 0] = rng.random(Nc) < 0.5
        if Kc > 1:
            Xc[:, 1:] = rng.standard_normal((Nc, Kc - 1))
    W = (rng.standard_normal((Kc, Ng)) * (rng.random((Kc, Ng)) < effect_frac)).astype(np.float32)
    b = (rng.standard_normal(Ng) * theta).astype(np.float32)
    sig = rng.uniform(0.5, 2.0, Ng).astype(np.float32)
    Z = Xc @ W + b[None, :] + sig[None, :] * rng.standard_normal((Nc, Ng)).astype(np.float32)
    Psi = expit(np.clip(Z, -9, 9))
    lam = rng.lognormal(0.0, 1.0, Ng) * depth
    size = rng.lognormal(0.0, 0.5, Nc)
    N = rng.poisson(size[:, None] * lam[None, :])
    eff = None
    if L == 2:
        c1 = rng.binomial(N, Psi)
        counts = [c1.astype(np.float32), (N - c1).astype(np.float32)]
    else:
        l1, l2, l3 = (rng.integers(50, 301, Ng) for _ in range(3))
        eff = se_efflen(l1, l2, l3)
        Le = eff[:, [0, 4, 5]].astype(np.float64)
        phi = np.stack([Psi * Le[:, 0], (1 - Psi) * Le[:, 1], np.ones_like(Psi) * Le[:, 2]], -1)
        phi /= phi.sum(-1, keepdims=True)
        c = rng.multinomial(N, phi)
        counts = [c[..., i].astype(np.float32) for i in range(3)]
    return dict(counts=counts, Xc=Xc, effLen=eff, W_true=W, b_true=b, sigma_true=sig,
                Psi_true=Psi.astype(np.float32))
