"""Pins for the simulator oracle (oracle/sim_oracle.c): the reference samples with unseeded TF / NumPy samplers
(brie/models/simulator.py:35,68-69), so what it fixes is the DISTRIBUTION -- checked here against scipy."""
import numpy as np
import pytest
from scipy import stats

from oracle import philox, sim_oracle


def _uniform53(gene, cell, draw, k, seed):
    w = [int(np.asarray(x).reshape(-1)[0]) for x in philox.philox4x32_10(
        np.uint32(gene), np.uint32(cell), np.uint32(draw), np.uint32(k),
        np.uint32(seed & 0xFFFFFFFF), np.uint32(seed >> 32))]
    return [((w[0] >> 5) * 67108864.0 + (w[1] >> 6) + 0.5) / 2.0 ** 53,
            ((w[2] >> 5) * 67108864.0 + (w[3] >> 6) + 0.5) / 2.0 ** 53]


@pytest.mark.parametrize("n,p", [(3, 0.2), (12, 0.5), (7, 0.9), (50, 0.05)])
def test_inversion_branch_against_a_python_walk_of_the_cdf(n, p):
    """Same uniform (first 53-bit value of the (gene, cell) stream) pushed through scipy's exact cdf."""
    for gene, cell in [(0, 0), (5, 17), (123456, 7)]:
        u = _uniform53(gene, cell, 0xFFFFFFFD, 0, 42)[0]
        pp = min(p, 1 - p)
        x = int(np.searchsorted(stats.binom.cdf(np.arange(n + 1), n, pp), u, side="left"))
        want = n - x if p > 0.5 else x
        assert sim_oracle.binomial(n, p, gene=gene, cell=cell, seed=42) == want


@pytest.mark.parametrize("n,p", [(5, 0.3), (40, 0.1), (100, 0.5), (1000, 0.02), (1000, 0.3), (30, 0.97), (20000, 0.41)])
def test_binomial_distribution(n, p):
    xs = np.array([sim_oracle.binomial(n, p, gene=g, cell=c, seed=7) for g in range(150) for c in range(100)])
    lo, hi = int(xs.min()), int(xs.max())
    obs = np.bincount(xs.astype(int) - lo, minlength=hi - lo + 1)
    ex = stats.binom.pmf(np.arange(lo, hi + 1), n, p) * len(xs)
    keep = ex > 5
    chi = ((obs[keep] - ex[keep]) ** 2 / ex[keep]).sum()
    assert stats.chi2.sf(chi, keep.sum() - 1) > 1e-4            # seeded: deterministic outcome
    assert abs(xs.mean() - n * p) < 5 * np.sqrt(n * p * (1 - p) / len(xs))


def test_multinomial_counts_conserve_depth_and_match_moments():
    rng = np.random.default_rng(0)
    Nc, Ng = 3000, 6
    psi = np.tile(rng.uniform(0.1, 0.9, (1, Ng)).astype(np.float32), (Nc, 1))
    total = np.tile(np.array([[1, 4, 11, 50, 400, 3000]], np.float32), (Nc, 1))
    eff = rng.integers(50, 300, (Ng, 6)).astype(np.float32)
    c1, c2, c3 = sim_oracle.simulate_counts(psi, total, eff, seed=3)
    np.testing.assert_array_equal(c1 + c2 + c3, total)
    w = np.stack([psi[0] * eff[:, 0], (1 - psi[0]) * eff[:, 4], eff[:, 5]])
    phi = w / w.sum(0)
    for k, c in enumerate((c1, c2, c3)):
        z = (c.mean(0) - total[0] * phi[k]) / np.sqrt(total[0] * phi[k] * (1 - phi[k]) / Nc)
        assert np.abs(z).max() < 4.5
    a, b, none = sim_oracle.simulate_counts(psi, total, None, seed=3)             # two categories
    assert none is None
    np.testing.assert_array_equal(a + b, total)
    part = sim_oracle.simulate_counts(psi[:, 4:], total[:, 4:], eff[4:], seed=3, gene_offset=4)
    np.testing.assert_array_equal(part[0], c1[:, 4:])                              # addressed by global gene


def test_psi_step_clips_and_uses_the_shared_stream():
    mean = np.zeros((50, 40), np.float32)
    psi = sim_oracle.simulate_psi(mean, np.full(40, 30.0, np.float32), seed=1)
    assert psi.min() >= 1 / (1 + np.exp(9.0)) * 0.999 and psi.max() <= 1 / (1 + np.exp(-9.0)) * 1.001
    z = np.log(sim_oracle.simulate_psi(mean, np.ones(40, np.float32), seed=1).astype(np.float64))
    z = z - np.log1p(-np.exp(z))
    np.testing.assert_allclose(z, philox.normal(1, sim_oracle.SIM_PSI_DRAW, 0, 50, 40), atol=1e-5)
