"""Count simulator (brie/models/simulator.py) on the GPU, through the C ABI, against oracle/sim_oracle.*:
integer outputs must be bit-identical, Psi within fp32 rounding."""
import numpy as np
import pytest

from oracle import sim_oracle
from tests.fakes import FakeAnnData

pytestmark = pytest.mark.gpu


def _inputs(Nc, Ng, seed, depth=3.0, with_eff=True):
    rng = np.random.default_rng(seed)
    psi = rng.beta(0.7, 0.7, (Nc, Ng)).astype(np.float32)
    total = rng.poisson(depth * np.exp(rng.normal(0, 1, (1, Ng))), (Nc, Ng)).astype(np.float32)
    eff = rng.integers(20, 400, (Ng, 6)).astype(np.float32) if with_eff else None
    return psi, total, eff


@pytest.mark.parametrize("Nc,Ng,depth,with_eff,gene_offset", [
    (200, 500, 3.0, True, 0),           # scRNA-like depths: the inversion branch
    (64, 257, 400.0, True, 1024),       # deep: BTRS branch, with a gene offset
    (33, 7, 30.0, False, 0),            # two categories (no effLen)
    (1, 1, 5000.0, True, 4),
])
def test_counts_bit_identical_to_oracle(lib, Nc, Ng, depth, with_eff, gene_offset):
    from brie_amd import _capi
    psi, total, eff = _inputs(Nc, Ng, 11, depth, with_eff)
    psi[0, 0] = 0.0                      # degenerate probabilities
    psi[-1, -1] = 1.0
    got = _capi.simulate_counts(psi, total, eff, seed=2024, gene_offset=gene_offset)
    want = sim_oracle.simulate_counts(psi, total, eff, seed=2024, gene_offset=gene_offset)
    for g, w in zip(got, want):
        if w is None:
            assert g is None
        else:
            np.testing.assert_array_equal(g, w)
    third = got[2] if with_eff else 0.0
    np.testing.assert_array_equal(got[0] + got[1] + third, np.floor(total))        # depth is conserved
    if not with_eff:
        assert got[0][0, 0] == 0 and got[1][-1, -1] == 0


def test_counts_follow_the_multinomial(lib):
    """Moments against the reference's sampling model (simulator.py:55-69): E[c_k] = n phi_k, Var = n phi_k (1-phi_k)."""
    from brie_amd import _capi
    Nc, Ng = 4000, 8
    rng = np.random.default_rng(3)
    psi = np.tile(rng.uniform(0.05, 0.95, (1, Ng)).astype(np.float32), (Nc, 1))
    total = np.tile(np.array([[1, 2, 5, 9, 20, 60, 300, 2000]], np.float32), (Nc, 1))
    eff = rng.integers(50, 300, (Ng, 6)).astype(np.float32)
    c = _capi.simulate_counts(psi, total, eff, seed=5)
    w = np.stack([psi[0] * eff[:, 0], (1 - psi[0]) * eff[:, 4], eff[:, 5]])
    phi = w / w.sum(0)
    for k in range(3):
        mean, var = total[0] * phi[k], total[0] * phi[k] * (1 - phi[k])
        z = (c[k].mean(0) - mean) / np.sqrt(var / Nc)
        assert np.abs(z).max() < 4.5, (k, z)
        np.testing.assert_allclose(c[k].var(0), var, rtol=0.12)
    cov01 = ((c[0] - c[0].mean(0)) * (c[1] - c[1].mean(0))).mean(0)                 # Cov = -n phi_0 phi_1
    np.testing.assert_allclose(cov01[3:], (-total[0] * phi[0] * phi[1])[3:], rtol=0.25)


def test_gene_shard_and_row_slabs_do_not_change_the_draws(lib, monkeypatch):
    from brie_amd import _capi
    psi, total, eff = _inputs(300, 64, 9)
    whole = _capi.simulate_counts(psi, total, eff, seed=1)
    part = _capi.simulate_counts(psi[:, 20:52], total[:, 20:52], eff[20:52], seed=1, gene_offset=20)
    monkeypatch.setenv("BRIE_SIM_SLAB_ELEMS", str(64 * 7))                          # 7 rows per slab
    slabbed = _capi.simulate_counts(psi, total, eff, seed=1)
    for k in range(3):
        np.testing.assert_array_equal(whole[k][:, 20:52], part[k])
        np.testing.assert_array_equal(whole[k], slabbed[k])
    other = _capi.simulate_counts(psi, total, eff, seed=2)
    assert (other[0] != whole[0]).mean() > 0.2


def test_psi_from_prior_matches_oracle(lib, monkeypatch):
    from brie_amd import _capi
    rng = np.random.default_rng(1)
    Nc, Ng = 120, 203
    mean = rng.normal(0, 3, (Nc, Ng)).astype(np.float32)
    sigma = rng.uniform(0.2, 6.0, Ng).astype(np.float32)          # wide enough to hit the +-9 clip
    got = _capi.simulate_psi(mean, sigma, seed=99, gene_offset=8)
    want = sim_oracle.simulate_psi(mean, sigma, seed=99, gene_offset=8)
    np.testing.assert_allclose(got, want, atol=2e-6)
    assert got.min() >= 1.0 / (1.0 + np.exp(9.0)) * 0.999 and got.max() <= 1.0
    monkeypatch.setenv("BRIE_SIM_SLAB_ELEMS", str(Ng * 11))
    np.testing.assert_array_equal(_capi.simulate_psi(mean, sigma, seed=99, gene_offset=8), got)


def test_simulator_api_on_a_fitted_object(lib):
    """fitBRIE -> simulator(mode='posterior' / 'prior') -> refit recovers the planted effect (power-analysis loop)."""
    import brie_amd
    from oracle.synth import make_problem
    Nc, Ng = 300, 40
    P = make_problem(Nc, Ng, Kc=1, L=3, seed=4)
    ad = FakeAnnData({'isoform1': P["counts"][0].copy(), 'isoform2': P["counts"][1].copy(),
                      'ambiguous': P["counts"][2].copy()}, effLen=P["effLen"])
    brie_amd.fitBRIE(ad, Xc=P["Xc"], min_iter=600, seed=1, verbose=False)
    ad.copy = lambda: FakeAnnData({k: np.array(v) for k, v in ad.layers.items()}, effLen=P["effLen"])
    depth = sum(ad.layers[k] for k in ('isoform1', 'isoform2', 'ambiguous'))

    sim = brie_amd.models.simulator(ad, seed=7)
    np.testing.assert_array_equal(ad.layers['Psi_sim'], ad.layers['Psi'])
    np.testing.assert_array_equal(sum(sim.layers[k] for k in ('isoform1', 'isoform2', 'ambiguous')), depth)
    want = sim_oracle.simulate_counts(ad.layers['Psi'], depth, P["effLen"], seed=7)
    np.testing.assert_array_equal(sim.layers['isoform1'], want[0])
    assert not np.array_equal(sim.layers['isoform1'], ad.layers['isoform1'])       # the input object keeps its reads

    sim2 = brie_amd.models.simulator(ad, mode="prior", seed=8)
    mean = P["Xc"] @ np.asarray(ad.varm['cell_coeff']).T + np.asarray(ad.varm['intercept']).T
    np.testing.assert_allclose(ad.layers['Psi_sim_noNoise'], 1 / (1 + np.exp(-mean)), atol=1e-6)
    np.testing.assert_allclose(ad.layers['Psi_sim'], sim_oracle.simulate_psi(
        mean.astype(np.float32), np.asarray(ad.varm['sigma']).reshape(-1), seed=8), atol=2e-6)
    assert set(sim2.layers) >= {'isoform1', 'isoform2', 'ambiguous'}

    with pytest.raises(ValueError):
        brie_amd.models.simulator(FakeAnnData({'isoform1': depth}), seed=1)         # no Psi
