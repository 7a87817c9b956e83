"""-m gpu: HIP path (through the C ABI) vs the CPU oracle on identical seeded inputs.

Tolerances (floating point path; north star: PSI within 1e-4 of the CPU path):
  * noise stream eps: 2e-6 absolute (fp32 Box-Muller vs fp64-rounded oracle)
  * a few Adam steps: assert_states_close -- 99.9 % of every state array within 1e-5,
    every element within 1e-3 except at most one sign-flipped element per array
    (Adam's first update of a fresh optimiser is +-lr whatever |g| is, so an
    element with g ~ 0 moves on the sign of a rounding error), itself bounded by
    2.2 lr per fresh optimiser; sized by 1 000 soak cases (profiles/history/r3c_soak_record.json).
  * PSI after a staged fit: tests/util.py::psi_null_rule -- HIP against the fp32 CPU oracle; genes
    that moved as a whole (displaced / clustered) and, in the quiet genes, entries beyond 1e-4
    are counted and bounded by what a SECOND fp32 CPU evaluation of the same algorithm (o32b:
    float Box-Muller, reversed sums, fused multiply-adds) does against that same oracle run --
    a direct fp32-vs-fp32 null (profiles/psi_null_r04.json).
"""
import numpy as np
import pytest

from oracle import philox
from tests import util

pytestmark = pytest.mark.gpu


_RECORD = None            # tests/tools/soak_randomised.py sets a list: (array, n, p99.9, max, n beyond `worst`) per call


def assert_states_close(so, sd, bulk=1e-5, worst=1e-3, lr=0.01, fresh=1):
    """State arrays of the oracle and of the device after a few Adam steps (the short-horizon parity rule), sized by
    what 1 000 soak cases of the two random families need (profiles/history/r3c_soak_record.json: 6 545 comparisons; the
    99.9 % quantile of an array of >= 1000 elements never above 1.9e-6; 6 comparisons with ONE element beyond 1e-3,
    the largest 2.3e-3) and 2 400 more in assert mode (profiles/history/r3f_soak_assert_mode.log, r3n_soak_wide.log: the
    largest 99.9 % quantile 7.6e-6, the Wc_loc of a ONE-cell problem).

      bulk    99.9 % of every array of >= 1000 elements within `bulk` (1e-5; smaller arrays fall under `worst` alone);
      worst   every element within `worst` (1e-3) -- EXCEPT sign flips: Keras Adam's first update of a fresh optimiser
              is lr * g / (|g| + 1e-7), i.e. +-lr whatever |g| is, so an element whose gradient is ~0 (zero coverage,
              mu on its prior mean: |g| ~ 1e-8) moves by +lr or -lr on the SIGN of a rounding error.  Such an element
              may sit up to 2 lr apart per fresh optimiser in the sequence (`fresh`; + 10 % for the steps that follow),
              and there may be at most max(1, 1e-4 n) of them per array."""
    if _RECORD is not None:
        for k, (n, p999, mx, n_out) in util.states_close_stats(so, sd, worst).items():
            _RECORD.append((k, n, p999, mx, n_out))
        return
    viol = util.states_close_violations(so, sd, bulk=bulk, worst=worst, lr=lr, fresh=fresh)
    assert not viol, viol


def test_library_loads(lib):
    assert lib.brie_abi_version() == 3


@pytest.mark.parametrize("gene_offset", [0, 1024])
def test_noise_stream_matches_oracle(lib, gene_offset):
    Nc, Ng, Kc = 37, 203, 2
    P = util.problem(Nc, Ng, Kc, 2)
    sh = util.device_shard(P, Nc, Ng, Kc, seed=0x1234567890AB, gene_offset=gene_offset)
    st = util.device_state(sh)
    D = philox.INIT_DRAW
    assert util.max_abs_diff(st["Z_loc"], philox.normal(0x1234567890AB, D, 0, Nc, Ng, gene_offset)) < 2e-6
    assert util.max_abs_diff(st["Z_std_log"], philox.normal(0x1234567890AB, D, 1, Nc, Ng, gene_offset)) < 2e-6
    assert util.max_abs_diff(st["Wc_loc"], philox.normal(0x1234567890AB, D, 2, Kc, Ng, gene_offset)) < 2e-6
    assert util.max_abs_diff(st["intercept"], philox.normal(0x1234567890AB, D, 3, 1, Ng, gene_offset)) < 2e-6
    assert np.all(st["sigma_log"] == 0)


CASES = [  # Nc, Ng, Kc, L, MC
    (64, 40, 0, 2, 1),
    (200, 500, 1, 2, 1),
    (150, 300, 3, 2, 3),
    (120, 260, 1, 3, 1),
    (90, 131, 2, 3, 3),
    (33, 7, 8, 2, 1),
]


@pytest.mark.parametrize("Nc,Ng,Kc,L,MC", CASES)
def test_single_steps_match_oracle(lib, Nc, Ng, Kc, L, MC):
    P = util.problem(Nc, Ng, Kc, L)
    seed = 99
    o = util.oracle_model(P, Nc, Ng, Kc, seed, np.float32)
    sh = util.device_shard(P, Nc, Ng, Kc, seed)
    tr_o = o.minimize(P["counts_pc"], P["Xc"], 3, 0.01, MC)
    tr_d = sh.step(3, 0.01, MC)
    np.testing.assert_allclose(tr_d, tr_o, rtol=2e-5)
    assert_states_close(util.oracle_state(o), util.device_state(sh))


def test_effLen_two_layers(lib):
    """effLen with only 2 count layers (model_TFProb.py:168-183 without 184-185)."""
    Nc, Ng, Kc = 70, 90, 1
    P = util.problem(Nc, Ng, Kc, 3)
    P["counts"] = P["counts"][:2]
    P["counts_pc"] = P["counts_pc"][:2]
    o = util.oracle_model(P, Nc, Ng, Kc, 5, np.float32)
    sh = util.device_shard(P, Nc, Ng, Kc, 5)
    tr_o = o.minimize(P["counts_pc"], P["Xc"], 2, 0.02, 1)
    tr_d = sh.step(2, 0.02, 1)
    np.testing.assert_allclose(tr_d, tr_o, rtol=2e-5)
    assert_states_close(util.oracle_state(o), util.device_state(sh))


def test_fixed_intercept_and_sigma(lib):
    Nc, Ng, Kc = 50, 64, 1
    P = util.problem(Nc, Ng, Kc, 2)
    o = util.oracle_model(P, Nc, Ng, Kc, 3, np.float32, intercept=0.0, sigma=1.5)
    sh = util.device_shard(P, Nc, Ng, Kc, 3, intercept=0.0, sigma=1.5)
    o.minimize(P["counts_pc"], P["Xc"], 5, 0.01, 1)
    sh.step(5, 0.01, 1)
    so, sd = util.oracle_state(o), util.device_state(sh)
    assert np.all(sd["intercept"] == 0.0)
    np.testing.assert_allclose(sd["sigma_log"], np.log(np.float32(1.5)), rtol=1e-6)
    assert_states_close(so, sd)


@pytest.mark.parametrize("Nc,Ng,Kc,L,MC,min_iter", [(200, 500, 1, 2, 1, 600), (100, 120, 1, 3, 3, 300)])
def test_psi_after_staged_fit(lib, Nc, Ng, Kc, L, MC, min_iter):
    """BASELINE metric 'PSI delta vs CPU ref' on config-1-sized problems."""
    from brie_amd import _capi
    P = util.problem(Nc, Ng, Kc, L, theta=3.0)
    seed = 11
    o32 = util.oracle_model(P, Nc, Ng, Kc, seed, np.float32)
    o32b = util.oracle_model(P, Nc, Ng, Kc, seed, np.float32, variant_b=True)     # the null: a second fp32 evaluation
    sh = util.device_shard(P, Nc, Ng, Kc, seed)
    for n, lr in util.staged_schedule(min_iter):
        for o in (o32, o32b):
            o.reset_optimizer()
            o.minimize(P["counts_pc"], P["Xc"], n, lr, MC)
        sh.reset_optimizer()
        sh.step(n, lr, MC)
    psi_d = sh.read(_capi.PSI)
    d_dev = np.abs(psi_d - o32.Psi)
    d_nul = np.abs(o32b.Psi - o32.Psi)
    print("PSI delta vs fp32 oracle: HIP max %.3g p99 %.3g | second fp32 CPU evaluation max %.3g p99 %.3g"
          % (d_dev.max(), np.percentile(d_dev, 99), d_nul.max(), np.percentile(d_nul, 99)))
    print(util.psi_null_of(sh, o32, o32b, what="Psi"))       # the parity rule, stated once (tests/util.py)
    # same yardstick for the interval width and the prior width: bulk tight, worst element bounded by what the second
    # fp32 evaluation does against the same oracle run
    for name, dev, ref, nul in (("Psi95CI", sh.read(_capi.PSI95CI), o32.Psi95CI, o32b.Psi95CI),
                                ("sigma", sh.read(_capi.SIGMA), o32.sigma, o32b.sigma)):
        d = np.abs(dev - ref)
        dn = np.abs(nul - ref)
        assert np.percentile(d, 99) <= max(2e-4, 3 * np.percentile(dn, 99)), (name, float(np.percentile(d, 99)))
        assert d.max() <= max(2e-3, 3 * dn.max()), (name, float(d.max()), float(dn.max()))


@pytest.mark.parametrize("Kc", [0, 1])
def test_psi_after_full_default_schedule(lib, Kc):
    """configs[0] (200 x 500) through the WHOLE BRIE2.fit default schedule (6 x 166 steps, fresh Adam per stage,
    model_TFProb.py:234-241), HIP vs the C restatement in fp32 (oracle/brie_oracle.c), judged by the second fp32 build
    of that restatement (-DBRIE_ORACLE_B) against the first."""
    from brie_amd import _capi
    from oracle.c_oracle import COracle
    Nc, Ng = 200, 500
    P = util.problem(Nc, Ng, Kc, 2, theta=3.0)
    seed = 11
    o32 = COracle(P["counts_pc"], P["Xc"], seed=seed, dtype=np.float32)
    o32b = COracle(P["counts_pc"], P["Xc"], seed=seed, dtype=np.float32, variant_b=True)
    o64 = COracle(P["counts_pc"], P["Xc"], seed=seed, dtype=np.float64)
    sh = util.device_shard(P, Nc, Ng, Kc, seed)
    for n, lr in util.staged_schedule(1000):
        for o in (o32, o32b, o64):
            o.reset_optimizer()
            o.minimize(n, lr, 1)
        sh.reset_optimizer()
        sh.step(n, lr, 1, trace=False)
    print("996 steps, Kc=%d:" % Kc, util.psi_null_of(sh, o32, o32b, what="Psi after 996 steps"))
    # ... and so that the fp32-vs-fp32 yardstick cannot drift: the same fit held against the
    # precision-independent answer -- the fp64 build of the restatement -- by round 3's rule (tests/util.py::psi_parity_rule)
    print("against the fp64 oracle:", util.psi_parity_of(sh, o32, o64, what="Psi after 996 steps vs fp64"))
    sh.close()


def test_loss_gene_matches_oracle(lib):
    Nc, Ng, Kc = 80, 100, 2
    P = util.problem(Nc, Ng, Kc, 2)
    o = util.oracle_model(P, Nc, Ng, Kc, 21, np.float64)
    sh = util.device_shard(P, Nc, Ng, Kc, 21)
    o.minimize(P["counts_pc"], P["Xc"], 2, 0.01, 1)
    sh.step(2, 0.01, 1)
    lg_o = o.eval_loss_gene(P["counts_pc"], P["Xc"], 25)
    lg_d = sh.loss_gene(25)
    np.testing.assert_allclose(lg_d, lg_o, rtol=1e-4, atol=1e-3)
    assert sh.draw == o.draw == 27


def test_gene_shard_invariance(lib):
    """Fitting a gene sub-block alone (gene_offset) == the same genes inside the full fit."""
    Nc, Ng, Kc = 60, 96, 1
    P = util.problem(Nc, Ng, Kc, 2)
    full = util.device_shard(P, Nc, Ng, Kc, 8)
    full.step(10, 0.01, 1)
    g0, g1 = 32, 80
    Ps = dict(P, counts=[c[:, g0:g1].copy() for c in P["counts"]])
    part = util.device_shard(Ps, Nc, g1 - g0, Kc, 8, gene_offset=g0)
    part.step(10, 0.01, 1)
    a, b = util.device_state(full), util.device_state(part)
    for k in util.STATE_KEYS:
        np.testing.assert_array_equal(a[k][:, g0:g1], b[k])


def test_errors_are_loud(lib):
    from brie_amd import _capi
    with pytest.raises(NotImplementedError):
        _capi.Shard(10, 10, Kg=1025)                                # more gene features than BRIE_MAX_KG_PANELS
    with pytest.raises(NotImplementedError):
        _capi.Shard(10, 12, intercept_mode=1, gene_offset=4)       # coupled modes cannot be gene-sharded
    with pytest.raises(ValueError):
        _capi.Shard(10, 10, gene_offset=3)
    sh = _capi.Shard(10, 12, 1)
    with pytest.raises(_capi.BrieError):
        sh.step(1, 0.01)                 # nothing uploaded yet
    with pytest.raises(ValueError):
        sh.upload(_capi.COUNT1, np.zeros((3, 3), np.float32))
    for bad_value in (-1.0, np.nan, np.inf):               # not counts: refused instead of NaN posteriors
        bad = _capi.Shard(6, 8)
        layer = np.ones((6, 8), np.float32)
        layer[3, 5] = bad_value
        bad.upload(_capi.COUNT1, layer)
        bad.upload(_capi.COUNT2, np.ones((6, 8), np.float32))
        bad.init_state()
        for _ in range(2):
            with pytest.raises(ValueError, match="negative or non-finite"):
                bad.step(1, 0.01, 1)
        bad.close()
    with pytest.raises(_capi.BrieError):                   # 2.9 TB of state: allocation fails, nothing leaks, no crash
        _capi.Shard(300000, 300000)
    ok = _capi.Shard(8, 8)                                 # the device is still usable afterwards, and the failed
    for l in range(2):                                     # hipMalloc is not reported again by a later call
        ok.upload(_capi.COUNT1 + l, np.ones((8, 8), np.float32))
    ok.init_state()
    assert np.isfinite(ok.step(2, 0.01, 1)).all()
    ok.close()


@pytest.mark.parametrize("Nc,Ng,Kc", [(1, 1, 0), (3, 5, 1), (2, 257, 0), (700, 4, 2), (5, 1030, 3)])
def test_ragged_and_tiny_shapes(lib, Nc, Ng, Kc):
    """Fewer rows than waves, gene counts that are not multiples of 4 / 256, one cell, one gene."""
    P = util.problem(Nc, Ng, Kc, 2, seed=3)
    o = util.oracle_model(P, Nc, Ng, Kc, 13, np.float32)
    sh = util.device_shard(P, Nc, Ng, Kc, 13)
    tr_o = o.minimize(P["counts_pc"], P["Xc"], 4, 0.02, 2)          # MC_size=2 -> generic (run-time MC) kernel
    tr_d = sh.step(4, 0.02, 2)
    np.testing.assert_allclose(tr_d, tr_o, rtol=5e-5, atol=1e-4)
    assert_states_close(util.oracle_state(o), util.device_state(sh))
    np.testing.assert_allclose(sh.loss_gene(3), o.eval_loss_gene(P["counts_pc"], P["Xc"], 3), rtol=1e-4, atol=1e-3)


def test_all_zero_counts_relax_to_prior(lib):
    """doc/brie_quant.rst:143-146: events without reads end at the prior (Psi 0.5 under a zero-mean prior)."""
    from brie_amd import _capi
    Nc, Ng = 40, 12
    P = {"counts": [np.zeros((Nc, Ng), np.float32)] * 2, "effLen": None, "Xc": np.zeros((Nc, 0), np.float32)}
    sh = util.device_shard(P, Nc, Ng, 0, 4, intercept=0.0, sigma=2.0)
    for n, lr in util.staged_schedule(900):
        sh.reset_optimizer()
        tr = sh.step(n, lr, 1)
    assert np.all(np.isfinite(tr))
    assert np.max(np.abs(sh.read(_capi.PSI) - 0.5)) < 0.02
    np.testing.assert_allclose(sh.read(_capi.Z_STD), 2.0, rtol=0.03)
    assert sh.read(_capi.PSI95CI).min() > 0.9


def test_large_counts_and_clip(lib):
    """Deep coverage drives Z_loc to the clip; everything stays finite and Psi -> c1/(c1+c2)."""
    from brie_amd import _capi
    Nc, Ng = 16, 8
    c1 = np.full((Nc, Ng), 5000, np.float32)
    c1[:, ::2] = 0
    c2 = 5000 - c1
    P = {"counts": [c1, c2], "effLen": None, "Xc": np.zeros((Nc, 0), np.float32)}
    sh = util.device_shard(P, Nc, Ng, 0, 6)
    o = util.oracle_model(P, Nc, Ng, 0, 6, np.float32)
    cnt = [c1 + np.float32(0.01), c2 + np.float32(0.01)]
    for n, lr in util.staged_schedule(1200):
        sh.reset_optimizer(); o.reset_optimizer()
        tr = sh.step(n, lr, 1)
        o.minimize(cnt, None, n, lr, 1)
    z = sh.read(_capi.Z_LOC)
    assert np.all(np.isfinite(tr)) and np.abs(z).max() <= 9.0
    psi = sh.read(_capi.PSI)
    assert psi[:, 1::2].min() > 0.99 and psi[:, ::2].max() < 0.01
    np.testing.assert_allclose(z, o.Z_loc, atol=5e-3)


@pytest.mark.parametrize("L,Kc,MC", [(2, 1, 1), (3, 2, 3), (2, 0, 2)])
def test_compact_u8_counts_bit_identical_to_fp32(lib, L, Kc, MC):
    """Integer counts <= 255 are stored as one byte per element and the pseudo-count of
    model_wrap.py:113-117 is applied in registers: every result must be bit-identical to the fp32 layers."""
    from brie_amd import _capi
    Nc, Ng = 90, 300
    P = util.problem(Nc, Ng, Kc, L, seed=19)
    assert max(c.max() for c in P["counts"]) <= 255
    a = util.device_shard(P, Nc, Ng, Kc, 23)
    b = util.device_shard(P, Nc, Ng, Kc, 23, storage="f32")
    tr_a, tr_b = a.step(7, 0.01, MC), b.step(7, 0.01, MC)
    assert a.count_storage == "u8" and b.count_storage == "f32"
    assert a.step_storage_bytes() == Nc * Ng * (48 + L) and b.step_storage_bytes() == Nc * Ng * (48 + 4 * L)
    assert a.step_algorithmic_bytes() == b.step_algorithmic_bytes() == Nc * Ng * (48 + 4 * L)
    np.testing.assert_array_equal(tr_a, tr_b)
    sa, sb = util.device_state(a), util.device_state(b)
    for k in util.STATE_KEYS:
        np.testing.assert_array_equal(sa[k], sb[k])
    np.testing.assert_array_equal(a.loss_gene(9), b.loss_gene(9))
    for l in range(L):                                       # read-back materialises the pseudo-count
        np.testing.assert_array_equal(a.read(_capi.COUNT1 + l), P["counts_pc"][l])
        np.testing.assert_array_equal(b.read(_capi.COUNT1 + l), P["counts_pc"][l])


def test_count_storage_tiers(lib):
    """Counts above 255 use u16 (for the gene quads that hold one; bit-identical to fp32), above 65535 or fractional
    keep fp32."""
    from brie_amd import _capi
    Nc, Ng, Kc = 40, 64, 1
    P = util.problem(Nc, Ng, Kc, 2, seed=29)
    big = dict(P, counts=[c.copy() for c in P["counts"]])
    big["counts"][0][3, 5] = 300.0
    big["counts"][1][11, 63] = 65535.0
    huge = dict(P, counts=[c.copy() for c in P["counts"]])
    huge["counts"][0][3, 5] = 70000.0
    frac = dict(P, counts=[c.copy() for c in P["counts"]])
    frac["counts"][1][7, 9] += 0.5
    allbig = dict(P, counts=[c.copy() for c in P["counts"]])
    allbig["counts"][0][3, ::4] = 300.0           # every gene quad holds a count > 255
    allbig["counts"][1][11, 63] = 65535.0
    for Q, want in ((big, "u8/u16 per gene quad"), (allbig, "u16"), (huge, "f32"), (frac, "f32")):
        Q["counts_pc"] = util.add_pseudo_count(Q["counts"], 0.01)
        sh = util.device_shard(Q, Nc, Ng, Kc, 31)
        ref = util.device_shard(Q, Nc, Ng, Kc, 31, storage="f32")
        o = util.oracle_model(Q, Nc, Ng, Kc, 31, np.float32)
        tr = sh.step(3, 0.01, 1)
        assert sh.count_storage == want
        np.testing.assert_array_equal(tr, ref.step(3, 0.01, 1))
        np.testing.assert_array_equal(sh.read(_capi.Z_LOC), ref.read(_capi.Z_LOC))
        np.testing.assert_array_equal(sh.read(_capi.COUNT1), Q["counts_pc"][0])
        np.testing.assert_allclose(tr, o.minimize(Q["counts_pc"], Q["Xc"], 3, 0.01, 1), rtol=2e-5)
    # no pseudo-count at all: the decision is taken at the first step
    sh = util.device_shard(P, Nc, Ng, Kc, 31, pseudo=0)
    o = util.oracle_model(P, Nc, Ng, Kc, 31, np.float32)
    tr = sh.step(3, 0.01, 1)
    assert sh.count_storage == "u8"
    np.testing.assert_allclose(tr, o.minimize(P["counts"], P["Xc"], 3, 0.01, 1), rtol=2e-5)


@pytest.mark.parametrize("L,MC,Kg", [(2, 1, 0), (3, 3, 0), (2, 1, 2), (2, 1, 6)])
def test_mixed_count_tiers_per_gene_quad(lib, L, MC, Kg):
    """Only some gene quads hold a count > 255: those keep 2 bytes per count, all others 1 (one launch, every lane
    picks up its quad's width and row offset).  Bit-identical to fp32 storage for steps, loss_gene, per-batch packing
    (which first re-tiers to one u16 tier) and the count read-back; close to the oracle."""
    from brie_amd import _capi
    Nc, Ng, Kc = 70, 1100, 2                # 5 gene blocks, the last partly filled
    P = util.problem(Nc, Ng, Kc, L, seed=41)
    P["counts"] = [c.copy() for c in P["counts"]]
    P["counts"][0][3, 300] = 999.0          # block 1
    P["counts"][1][69, 1099] = 40000.0      # block 4 (the ragged one), its last quad
    P["counts"][0][0, 0] = 300.0            # first lane of block 0
    P["counts"][1][5, 255] = 256.0          # last lane of block 0 ...
    P["counts"][L - 1][7, 257] = 65535.0    # ... and its neighbour, the first lane of block 1
    P["counts"][0][11, 702] = 1000.0
    hot = np.unique(np.concatenate([np.nonzero((c > 255).any(axis=0))[0] // 4 for c in P["counts"]]))
    assert hot.tolist() == [0, 63, 64, 75, 175, 274]
    P["counts_pc"] = util.add_pseudo_count(P["counts"], 0.01)
    if Kg:
        P["Xg"] = np.random.default_rng(3).normal(size=(Ng, Kg)).astype(np.float32)
    sh = util.device_shard(P, Nc, Ng, Kc, 31, Kg=Kg)
    ref = util.device_shard(P, Nc, Ng, Kc, 31, storage="f32", Kg=Kg)
    tr = sh.step(4, 0.01, MC)
    assert sh.count_storage == "u8/u16 per gene quad"
    assert sh.step_storage_bytes() == Nc * (Ng * 48 + L * (Ng + 4 * len(hot)))
    np.testing.assert_array_equal(tr, ref.step(4, 0.01, MC))
    # (the forward-only pass is a different template instantiation per storage: fused multiply-adds may be
    # contracted differently, so its sums agree to the last ulp or two, not bit for bit)
    np.testing.assert_allclose(sh.loss_gene(5), ref.loss_gene(5), rtol=5e-7)
    for l in range(L):
        np.testing.assert_array_equal(sh.read(_capi.COUNT1 + l), P["counts_pc"][l])
    if Kg == 0:
        o = util.oracle_model(P, Nc, Ng, Kc, 31, np.float32)
        np.testing.assert_allclose(tr, o.minimize(P["counts_pc"], P["Xc"], 4, 0.01, MC), rtol=2e-5)
        # freeze most genes: packing permutes quads across gene blocks -> one u16 tier, still bit-identical
        mask = np.zeros(Ng, bool)
        mask[290:310] = True
        mask[1090:] = True
        for s_ in (sh, ref):
            s_.set_gene_mask(mask)
        np.testing.assert_array_equal(sh.step(3, 0.005, MC), ref.step(3, 0.005, MC))
        assert sh.count_storage == "u16"
        for s_ in (sh, ref):
            s_.set_gene_mask(None)
    np.testing.assert_array_equal(sh.step(2, 0.005, MC), ref.step(2, 0.005, MC))
    for k, a in util.device_state(sh).items():
        np.testing.assert_array_equal(a, util.device_state(ref)[k], err_msg=k)
    sh.close()
    ref.close()


@pytest.mark.parametrize("mode,Kg,Kc,L,MC", [("cell", 0, 1, 2, 1), ("gene", 2, 1, 2, 1), ("cell", 4, 2, 3, 3),
                                             ("gene", 1, 0, 2, 2), ("cell", 3, 0, 2, 1),
                                             # Kg > 4: Xg tile in LDS, Wg_loc row broadcast with v_readlane
                                             ("gene", 5, 1, 2, 1), ("gene", 8, 3, 2, 1), ("cell", 16, 2, 3, 2), ("gene", 33, 8, 2, 3),
                                             ("gene", 64, 0, 2, 1),
                                             # wide cell design (Kc > 8) together with the coupled terms
                                             ("cell", 0, 12, 2, 1), ("gene", 3, 20, 3, 2), ("cell", 9, 33, 2, 1)])
def test_coupled_modes_match_oracle(lib, mode, Kg, Kc, L, MC):
    """Gene features Xg with per-cell weights Wg_loc (model_TFProb.py:124-125) and per-cell intercept /
    sigma (intercept_mode='cell', :53-55): per-cell statistics are wave-reduced over genes on the device."""
    from brie_amd import _capi
    Nc, Ng = 150, 600            # 3 gene blocks (the last one partly filled), rows split over several chunks
    P = util.problem(Nc, Ng, Kc, L, seed=37)
    P["Xg"] = np.random.default_rng(5).standard_normal((Ng, Kg)).astype(np.float32)
    o = util.oracle_model(P, Nc, Ng, Kc, 41, np.float32, Kg=Kg, mode=mode)
    sh = util.device_shard(P, Nc, Ng, Kc, 41, Kg=Kg, mode=mode)
    s0 = util.device_state(sh)
    for k in util.STATE_KEYS:                                       # Model_init incl. Wg_loc / per-cell intercept
        assert util.max_abs_diff(s0[k], getattr(o, k)) < 2e-6, k
    tr_o = o.minimize(P["counts_pc"], P["Xc"], 6, 0.01, MC)
    tr_d = sh.step(6, 0.01, MC)
    np.testing.assert_allclose(tr_d, tr_o, rtol=3e-5)
    assert_states_close(util.oracle_state(o), util.device_state(sh))
    par = (Nc, 1) if mode == "cell" else (1, Ng)
    assert sh.read(_capi.INTERCEPT).shape == par and sh.read(_capi.SIGMA).shape == par
    np.testing.assert_allclose(sh.read(_capi.SIGMA), o.sigma, rtol=1e-4)
    np.testing.assert_allclose(sh.loss_gene(5), o.eval_loss_gene(P["counts_pc"], P["Xc"], 5), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("Kg", [2, 9, 70])         # registers / LDS tile / 64-feature panels, through BRIE2.fit
def test_coupled_fit_through_python_api(lib, Kg):
    import brie_amd
    Nc, Ng, Kc = 80, 70, 1
    P = util.problem(Nc, Ng, Kc, 2, seed=43)
    Xg = (np.random.default_rng(6).standard_normal((Ng, Kg)) * (0.3 if Kg > 64 else 1.0)).astype(np.float32)
    m = brie_amd.BRIE2(Nc, Ng, Kc=Kc, Kg=Kg, intercept_mode='cell', seed=9)
    losses = m.fit(P["counts"], Xc=P["Xc"], Xg=Xg, min_iter=120, max_iter=120, n_loss_gene=5, pseudo_count=0.01,
                   verbose=False)
    o = util.oracle_model(dict(P, Xg=Xg), Nc, Ng, Kc, 9, np.float64, Kg=Kg, mode='cell')
    lo = o.fit(P["counts_pc"], P["Xc"], min_iter=120, max_iter=120, n_loss_gene=5)
    np.testing.assert_allclose(losses.numpy(), lo, rtol=2e-4)
    assert m.Wg_loc.shape == (Nc, Kg) and m.intercept.shape == (Nc, 1) and m.sigma.shape == (Nc, 1)
    np.testing.assert_allclose(m.Wg_loc.numpy(), o.Wg_loc, atol=2e-2)
    np.testing.assert_allclose(m.intercept.numpy(), o.intercept, atol=2e-2)      # fp32 vs fp64, 120 noisy steps
    rv = brie_amd.BRIE_RV(m)
    assert rv.gene_coeff.shape == (Nc, Kg) and rv.intercept_mode == 'cell'


@pytest.mark.parametrize("Nc,Ng,Kc,L,MC", [(90, 300, 2, 2, 1), (60, 140, 1, 3, 4), (33, 9, 0, 2, 3)])
def test_marginlik_target_matches_oracle(lib, Nc, Ng, Kc, L, MC):
    """target="marginLik" (model_TFProb.py:156-157,188-189,202-205): prior samples, log-mean-exp, only the
    prior parameters move; the posterior arrays must stay bit-identical."""
    P = util.problem(Nc, Ng, Kc, L, seed=47)
    o = util.oracle_model(P, Nc, Ng, Kc, 53, np.float32)
    sh = util.device_shard(P, Nc, Ng, Kc, 53)
    sh.set_target("marginLik")
    s0 = util.device_state(sh)
    tr_o = o.minimize(P["counts_pc"], P["Xc"], 6, 0.02, MC, target="marginLik")
    tr_d = sh.step(6, 0.02, MC)
    np.testing.assert_allclose(tr_d, tr_o, rtol=3e-5)
    s1 = util.device_state(sh)
    np.testing.assert_array_equal(s1["Z_loc"], s0["Z_loc"])
    np.testing.assert_array_equal(s1["Z_std_log"], s0["Z_std_log"])
    assert_states_close(util.oracle_state(o), s1)
    lg_o = o.eval_loss_gene(P["counts_pc"], P["Xc"], 7, target="marginLik")
    np.testing.assert_allclose(sh.loss_gene(7), lg_o, rtol=1e-4, atol=1e-3)
    sh.set_target("ELBO")                                          # and back
    np.testing.assert_allclose(sh.step(2, 0.01, 1), o.minimize(P["counts_pc"], P["Xc"], 2, 0.01, 1), rtol=3e-5)


@pytest.mark.parametrize("mode,Kg,Kc,L,MC", [("gene", 2, 1, 2, 3), ("cell", 0, 2, 3, 2), ("cell", 3, 0, 2, 1),
                                             ("gene", 9, 1, 2, 4), ("gene", 0, 12, 2, 3), ("cell", 6, 20, 3, 2)])
def test_marginlik_target_coupled_and_wide_models(lib, mode, Kg, Kc, L, MC):
    """target="marginLik" with gene features, per-cell intercept / sigma and wide cell designs (the reference has no
    such restriction, model_TFProb.py:156-157,188-189,202-205): the MARGIN variants of the step kernel."""
    from brie_amd import _capi
    Nc, Ng = 150, 600
    P = util.problem(Nc, Ng, Kc, L, seed=61)
    P["Xg"] = np.random.default_rng(7).normal(size=(Ng, Kg)).astype(np.float32) * 0.5
    o = util.oracle_model(P, Nc, Ng, Kc, 67, np.float32, Kg=Kg, mode=mode)
    sh = util.device_shard(P, Nc, Ng, Kc, 67, Kg=Kg, mode=mode)
    sh.set_target("marginLik")
    s0 = util.device_state(sh)
    tr_o = o.minimize(P["counts_pc"], P["Xc"], 5, 0.02, MC, target="marginLik")
    tr_d = sh.step(5, 0.02, MC)
    np.testing.assert_allclose(tr_d, tr_o, rtol=5e-5)
    s1 = util.device_state(sh)
    np.testing.assert_array_equal(s1["Z_loc"], s0["Z_loc"])               # the posterior is not touched
    np.testing.assert_array_equal(s1["Z_std_log"], s0["Z_std_log"])
    assert_states_close(util.oracle_state(o), s1, bulk=5e-5)
    lg_o = o.eval_loss_gene(P["counts_pc"], P["Xc"], 6, target="marginLik")
    np.testing.assert_allclose(sh.loss_gene(6), lg_o, rtol=2e-4, atol=2e-3)
    sh.set_target("ELBO")
    np.testing.assert_allclose(sh.step(2, 0.01, 1), o.minimize(P["counts_pc"], P["Xc"], 2, 0.01, 1), rtol=5e-5)
    sh.close()


@pytest.mark.parametrize("target,mode,Kg,Kc,L", [("ELBO", "gene", 0, 1, 2), ("ELBO", "gene", 0, 1, 3),
                                                 ("marginLik", "gene", 0, 2, 2), ("marginLik", "cell", 3, 1, 3),
                                                 ("marginLik", "gene", 0, 10, 2)])
def test_loglik_mc_accessor_matches_oracle(lib, target, mode, Kg, Kc, L):
    """BRIE2.logLik_MC (model_TFProb.py:130-191) per (cell, gene): mean over posterior samples (ELBO) or log-mean-exp
    over prior samples (marginLik), against the oracle's per-sample log-likelihood on the same noise."""
    from oracle import philox
    Nc, Ng, size = 40, 70, 5
    P = util.problem(Nc, Ng, Kc, L, seed=71)
    P["Xg"] = np.random.default_rng(9).normal(size=(Ng, Kg)).astype(np.float32) * 0.5
    o = util.oracle_model(P, Nc, Ng, Kc, 73, np.float64, Kg=Kg, mode=mode)
    sh = util.device_shard(P, Nc, Ng, Kc, 73, Kg=Kg, mode=mode)
    o.minimize(P["counts_pc"], P["Xc"], 3, 0.02, 1)
    sh.step(3, 0.02, 1)
    for k in util.STATE_KEYS:                      # same state on both sides: this test is about the accessor
        setattr(o, k, np.asarray(util.device_state(sh)[k], np.float64))
    sh.set_target(target)
    draw = sh.draw
    got = sh.loglik_mc(size)
    assert sh.draw == draw + 1
    eps = np.stack([philox.normal(73, draw, k, Nc, Ng) for k in range(size)]).astype(np.float64)
    cnt = [np.asarray(c, np.float64) for c in P["counts_pc"]]
    if target == "ELBO":
        z = o.Z_loc[None] + o.Z_std[None] * eps
        want = np.mean([o.loglik_terms(cnt, z[k])[0] for k in range(size)], axis=0)
    else:
        z = o.prior_mean(P["Xc"])[None] + np.exp(o.sigma_log)[None] * eps
        ll = np.stack([o.loglik_terms(cnt, z[k])[0] for k in range(size)])
        want = ll.max(0) + np.log(np.exp(ll - ll.max(0)).mean(0))
    np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-4)
    sh.close()


@pytest.mark.parametrize("Ng,L", [(1000, 2), (515, 3), (4, 2)])
def test_staged_host_ingest_is_bit_identical_to_the_plain_copy(lib, monkeypatch, Ng, L):
    """brie_upload of a pageable host count layer through the staged pipeline (host threads convert row slabs to u16 in
    page-locked buffers, asynchronous copies, a kernel writes the tiled layer; include/brie_amd.h) must leave exactly
    the layer the plain strided copy leaves: integer slabs travel as u16, a slab with a fractional / huge / negative-zero
    value as fp32, ragged last slabs, Ng not a multiple of 4."""
    from brie_amd import _capi
    Nc, Kc = 333, 1
    P = util.problem(Nc, Ng, Kc, L, seed=21)
    cnt = [np.array(c) for c in P["counts"]]
    cnt[1][8, 2 % Ng] = 300.0                      # > 255: a u16 quad next to u8 quads (the storage tiers still apply)

    def build(mode, slab_rows=None, frac=False):
        monkeypatch.setenv("BRIE_INGEST", mode)
        monkeypatch.setenv("BRIE_INGEST_THREADS", "3")
        if slab_rows:
            monkeypatch.setenv("BRIE_INGEST_SLAB_ELEMS", str(slab_rows * Ng))
        layers = [c.copy() for c in cnt]
        if frac:
            layers[0][7, 1 % Ng] = 70000.0         # > 65535: this slab travels as fp32 (and the shard stays fp32 storage)
            layers[0][200, 0] = 2.5                # fractional: its slab too, the others as u16
            layers[1][301, Ng - 1] = -0.0 if Ng > 1 else 0.0
        sh = _capi.Shard(Nc, Ng, Kc, n_layers=L, has_efflen=L == 3, seed=9)
        for l in range(L):
            sh.upload(_capi.COUNT1 + l, layers[l])
        got = [sh.read(_capi.COUNT1 + l) for l in range(L)]
        sh.add_pseudo_count(0.01)
        if L == 3:
            sh.upload(_capi.EFFLEN, P["effLen"])
        sh.upload(_capi.XC, P["Xc"])
        sh.init_state()
        tr = sh.step(5, 0.01, 1)
        out = (got, tr, sh.read(_capi.Z_LOC), sh.count_storage)
        sh.close()
        return out, layers
    for frac in (False, True):
        (ref, tr_ref, z_ref, st_ref), layers = build("direct", frac=frac)
        for l in range(L):
            np.testing.assert_array_equal(ref[l], layers[l])
        for slab_rows in (50, 64, 400):            # ragged last slab / exact multiple-ish / one slab
            (got, tr, z, st), _ = build("staged", slab_rows, frac=frac)
            for l in range(L):
                assert np.array_equal(got[l], ref[l]) and np.array_equal(np.signbit(got[l]), np.signbit(ref[l])), (l, slab_rows)
            np.testing.assert_array_equal(tr, tr_ref)
            np.testing.assert_array_equal(z, z_ref)
            assert st == st_ref


@pytest.mark.parametrize("target,mode,Kg,Kc,L,MC", [("marginLik", "gene", 0, 2, 2, 3), ("marginLik", "cell", 3, 1, 3, 4),
                                                    ("marginLik", "gene", 0, 10, 2, 3), ("marginLik", "gene", 6, 0, 2, 2),
                                                    ("ELBO", "gene", 0, 2, 2, 3), ("ELBO", "cell", 2, 1, 3, 1),
                                                    ("ELBO", "gene", 0, 12, 2, 2)])
def test_get_loss_per_gene_and_per_cell_match_oracle(lib, target, mode, Kg, Kc, L, MC):
    """brie_get_loss = BRIE2.get_loss(target, axis, MC_size) (model_TFProb.py:194-211) as one evaluation:
    target marginLik with MC_size > 1 is -sum log-mean-exp over the samples (:202-205) -- against the oracle's
    margin_loss_and_grads(need_grads=False) on the same noise; axis=1 sums the (Nc, Ng) terms over genes -- against the
    oracle's per-entry KL / log-likelihood summed the same way; the two axes of one evaluation have the same total."""
    from oracle import philox
    Nc, Ng = 40, 70
    P = util.problem(Nc, Ng, Kc, L, seed=71)
    P["Xg"] = np.random.default_rng(9).normal(size=(Ng, Kg)).astype(np.float32) * 0.5
    o = util.oracle_model(P, Nc, Ng, Kc, 73, np.float64, Kg=Kg, mode=mode)
    sh = util.device_shard(P, Nc, Ng, Kc, 73, Kg=Kg, mode=mode)
    o.minimize(P["counts_pc"], P["Xc"], 3, 0.02, 1)
    sh.step(3, 0.02, 1)
    for k in util.STATE_KEYS:                      # same state on both sides: this test is about the accessor
        setattr(o, k, np.asarray(util.device_state(sh)[k], np.float64))
    sh.set_target(target)
    cnt = [np.asarray(c, np.float64) for c in P["counts_pc"]]
    draw = sh.draw
    per_gene = sh.get_loss(MC, 0)
    used = sh.draw - draw
    assert used == (MC if target == "ELBO" else 1)           # ELBO: MC draw ids; marginLik: the MC samples of one
    sh.draw = draw
    per_cell = sh.get_loss(MC, 1)
    assert per_gene.shape == (Ng,) and per_cell.shape == (Nc,)
    if target == "marginLik":
        o.draw = draw
        want = o.margin_loss_and_grads(cnt, P["Xc"], MC, need_grads=False)
        np.testing.assert_allclose(per_gene, want["loss_gene"], rtol=2e-4, atol=2e-3)
        eps = np.stack([philox.normal(73, draw, k, Nc, Ng) for k in range(MC)]).astype(np.float64)
        z = o.prior_mean(P["Xc"])[None] + np.exp(o.sigma_log)[None] * eps
        ll = np.stack([o.loglik_terms(cnt, z[k])[0] for k in range(MC)])
        terms = -(ll.max(0) + np.log(np.exp(ll - ll.max(0)).mean(0)))
    else:
        eps = np.stack([philox.normal(73, draw + k, 0, Nc, Ng) for k in range(MC)]).astype(np.float64)
        z = o.Z_loc[None] + o.Z_std[None] * eps
        ll = np.mean([o.loglik_terms(cnt, z[k])[0] for k in range(MC)], axis=0)
        d, dl = o.Z_loc - o.prior_mean(P["Xc"]), o.Z_std_log - o.sigma_log
        terms = 0.5 * d * d * np.exp(-2 * o.sigma_log) + 0.5 * np.expm1(2 * dl) - dl - ll
        # ... and the fast pass (brie_loss_gene: the same samples in registers) gives the same per-gene values
        sh.draw = draw
        np.testing.assert_allclose(sh.loss_gene(MC), per_gene, rtol=2e-4, atol=2e-3)
    np.testing.assert_allclose(per_gene, terms.sum(0), rtol=2e-4, atol=2e-3)
    np.testing.assert_allclose(per_cell, terms.sum(1), rtol=2e-4, atol=2e-3)
    np.testing.assert_allclose(per_gene.astype(np.float64).sum(), per_cell.astype(np.float64).sum(), rtol=1e-5)
    sh.close()


@pytest.mark.parametrize("Kc,L,MC", [(9, 2, 1), (20, 3, 3), (33, 2, 2)])
def test_wide_cell_design_matches_oracle(lib, Kc, L, MC):
    """Kc > 8: Wc_loc tile in LDS for Xc.Wc_loc, Xc^T.r reduced by a hand-written v_mfma_f32_32x32x2_f32 kernel."""
    Nc, Ng = 300, 520
    P = util.problem(Nc, Ng, Kc, L, seed=57)
    o = util.oracle_model(P, Nc, Ng, Kc, 59, np.float32)
    sh = util.device_shard(P, Nc, Ng, Kc, 59)
    tr_o = o.minimize(P["counts_pc"], P["Xc"], 5, 0.01, MC)
    tr_d = sh.step(5, 0.01, MC)
    np.testing.assert_allclose(tr_d, tr_o, rtol=3e-5)
    assert_states_close(util.oracle_state(o), util.device_state(sh), bulk=5e-5)
    np.testing.assert_allclose(sh.loss_gene(5), o.eval_loss_gene(P["counts_pc"], P["Xc"], 5), rtol=1e-4, atol=1e-3)
    # the MFMA tile kernel keeps the residual on chip: a wide design moves no more HBM bytes than a narrow one
    assert sh.step_storage_bytes() <= sh.step_algorithmic_bytes()


@pytest.mark.parametrize("Kc,L,MC,mode,Kg", [(96, 2, 1, "gene", 0), (130, 3, 2, "gene", 0), (70, 2, 1, "cell", 2),
                                             (300, 2, 1, "gene", 0), (257, 2, 1, "gene", 0)])
def test_very_wide_cell_design_runs_in_panels(lib, Kc, L, MC, mode, Kg):
    """Kc > 64 (the reference has no limit, model_TFProb.py:84,122-123): Xc.Wc_loc is formed beside the
    streaming kernel (round 5: one tiled MFMA launch over all features in 32-feature stages) and Xc^T.r reduced from its
    residual (one launch per 256 features: a ragged last stage at Kc = 130; two launches at Kc = 300, the second through the
    narrower kernel; ONE feature in the second launch at Kc = 257; with gene features and per-cell intercepts next to it).
    Same oracle, same bounds as the one-pass wide designs."""
    from brie_amd import _capi
    Nc, Ng = 300, 520
    P = util.problem(Nc, Ng, Kc, L, seed=157)
    P["Xc"] = (P["Xc"] * (0.2 if Kc < 200 else 0.1)).astype(np.float32)   # many N(0,1) features: keep the prior mean inside the clip range
    if Kg:
        P["Xg"] = np.random.default_rng(9).standard_normal((Ng, Kg)).astype(np.float32)
    o = util.oracle_model(P, Nc, Ng, Kc, 59, np.float32, Kg=Kg, mode=mode)
    sh = util.device_shard(P, Nc, Ng, Kc, 59, Kg=Kg, mode=mode)
    tr_o = o.minimize(P["counts_pc"], P["Xc"], 5, 0.01, MC)
    tr_d = sh.step(5, 0.01, MC)
    np.testing.assert_allclose(tr_d, tr_o, rtol=3e-5)
    assert_states_close(util.oracle_state(o), util.device_state(sh), bulk=5e-5)
    np.testing.assert_allclose(sh.loss_gene(5), o.eval_loss_gene(P["counts_pc"], P["Xc"], 5), rtol=1e-4, atol=1e-3)
    assert sh.step_storage_bytes() > sh.step_algorithmic_bytes()          # the panels exchange through HBM, and say so
    if not Kg and mode == "gene":                          # the other target on the same design
        sh.set_target("marginLik")
        o.reset_optimizer()
        sh.reset_optimizer()
        tr_o = o.minimize(P["counts_pc"], P["Xc"], 3, 0.01, 3, target="marginLik")
        np.testing.assert_allclose(sh.step(3, 0.01, 3), tr_o, rtol=1e-4)
        d = np.abs(sh.read(_capi.WC_LOC).astype(np.float64) - o.Wc_loc)
        if Kc < 257:                                       # the designs that held this before round 5 keep it
            assert d.max() <= 2e-4, float(d.max())
        else:
            # a sign event of Keras Adam's first steps (+-lr whatever |g| is): with 257 x 520 weights ONE entry of 133 640 sat
            # 6.9e-4 off after the three steps (call r5y).  At most two such entries, none beyond 2 lr, and each must BE a
            # sign event: a weight whose gradient the oracle itself sees near zero (|m| of its Adam slot far below the bulk)
            out = np.argwhere(d > 2e-4)
            assert len(out) <= 2 and d.max() <= 2 * 0.01, (len(out), float(d.max()))
            m_abs = np.abs(np.asarray(o.slots["Wc_loc"].m, np.float64))
            for k, j in out:
                assert m_abs[k, j] <= 0.05 * np.median(m_abs), (int(k), int(j), float(m_abs[k, j]), float(np.median(m_abs)))
    with pytest.raises(NotImplementedError):
        _capi.Shard(Nc, Ng, 1025)
    sh.close()


@pytest.mark.parametrize("Kg,Kc,L,MC,mode", [(70, 0, 2, 1, "gene"), (96, 3, 2, 2, "cell"), (130, 20, 3, 1, "gene"),
                                             (65, 70, 2, 1, "cell"), (200, 130, 2, 1, "gene")])
def test_very_wide_gene_design_runs_in_panels(lib, Kg, Kc, L, MC, mode):
    """Kg > 64 (the reference has no limit, model_TFProb.py:85,124-125): Wg_loc.Xg^T joins the prior mean panel by panel,
    the Wg_loc gradient r.Xg is formed from the step's residual panel by panel (a ragged last panel at Kg = 130 and 65;
    next to no / a narrow / a wide / a very wide cell design; per-cell intercepts).  Same oracle and bounds as Kg <= 64."""
    from brie_amd import _capi
    Nc, Ng = 150, 600
    P = util.problem(Nc, Ng, Kc, L, seed=163)
    if Kc > 8:
        P["Xc"] = (P["Xc"] * 0.2).astype(np.float32)
    P["Xg"] = (np.random.default_rng(11).standard_normal((Ng, Kg)) * 0.3).astype(np.float32)
    o = util.oracle_model(P, Nc, Ng, Kc, 71, np.float32, Kg=Kg, mode=mode)
    sh = util.device_shard(P, Nc, Ng, Kc, 71, Kg=Kg, mode=mode)
    s0 = util.device_state(sh)
    for k in util.STATE_KEYS:
        assert util.max_abs_diff(s0[k], getattr(o, k)) < 2e-6, k
    tr_o = o.minimize(P["counts_pc"], P["Xc"], 6, 0.01, MC)
    tr_d = sh.step(6, 0.01, MC)
    np.testing.assert_allclose(tr_d, tr_o, rtol=3e-5)
    assert_states_close(util.oracle_state(o), util.device_state(sh), bulk=5e-5)
    np.testing.assert_allclose(sh.loss_gene(5), o.eval_loss_gene(P["counts_pc"], P["Xc"], 5), rtol=1e-4, atol=1e-3)
    assert sh.step_storage_bytes() > sh.step_algorithmic_bytes()
    if Kg == 70:                                           # the other target, and the per-entry accessor's prior mean
        sh.set_target("marginLik")
        o.reset_optimizer()
        sh.reset_optimizer()
        tr_o = o.minimize(P["counts_pc"], P["Xc"], 3, 0.01, 3, target="marginLik")
        np.testing.assert_allclose(sh.step(3, 0.01, 3), tr_o, rtol=1e-4)
        np.testing.assert_allclose(sh.read(_capi.WG_LOC), o.Wg_loc, atol=2e-4)
    sh.close()


def test_sparse_layers_densified_on_device(lib):
    """CSC / CSR / COO count layers (model_wrap.py:108-111 densifies on the host) incl. duplicate entries."""
    import scipy.sparse as sp
    from brie_amd import _capi
    Nc, Ng, Kc = 70, 530, 1
    P = util.problem(Nc, Ng, Kc, 2, seed=61)
    dense = util.device_shard(P, Nc, Ng, Kc, 67)
    for conv in (sp.csc_matrix, sp.csr_matrix, sp.coo_matrix):
        Q = dict(P, counts=[conv(c) for c in P["counts"]])
        sh = util.device_shard(Q, Nc, Ng, Kc, 67)
        np.testing.assert_array_equal(sh.read(_capi.COUNT1), P["counts_pc"][0])
        np.testing.assert_array_equal(sh.step(3, 0.01, 1), util.device_shard(P, Nc, Ng, Kc, 67).step(3, 0.01, 1))
    # duplicates are summed like scipy's toarray()
    rows, cols, vals = np.array([0, 0, 5, 5]), np.array([3, 3, 7, 7]), np.array([1, 2, 4, 4], np.float32)
    dup = sp.csc_matrix((vals, (rows, cols)), shape=(Nc, Ng))
    dup.has_canonical_format = False
    raw = sp.csc_matrix((Nc, Ng), dtype=np.float32)
    raw.indptr, raw.indices, raw.data = np.zeros(Ng + 1, np.int32), np.array([0, 0, 5, 5], np.int32), vals
    raw.indptr[4:] = 2
    raw.indptr[8:] = 4
    sh = _capi.Shard(Nc, Ng, 0)
    sh.upload(_capi.COUNT1, raw)
    sh.upload(_capi.COUNT2, np.zeros((Nc, Ng), np.float32))
    got = sh.read(_capi.COUNT1)
    assert got[0, 3] == 3 and got[5, 7] == 8 and got.sum() == 11
    dense.close()


def test_frozen_genes_and_loss_window(lib):
    """Per-batch convergence machinery: a gene mask freezes state, moments and per-gene parameters of the
    masked genes (their last loss is carried forward), active genes are unaffected; fully frozen 256-gene
    blocks are skipped.  Checked against the oracle with the same mask."""
    from brie_amd import _capi
    Nc, Ng, Kc = 60, 600, 1                      # 3 gene blocks; block 1 gets fully frozen
    P = util.problem(Nc, Ng, Kc, 2, seed=71)
    o = util.oracle_model(P, Nc, Ng, Kc, 73, np.float32)
    sh = util.device_shard(P, Nc, Ng, Kc, 73)
    o.minimize(P["counts_pc"], P["Xc"], 4, 0.01, 1)
    sh.step(4, 0.01, 1)
    win = sh.read_loss_window(4)
    np.testing.assert_allclose(win, np.asarray(o.lg_hist[-4:]), rtol=2e-5, atol=1e-3)
    mask = np.ones(Ng, bool)
    mask[256:512] = False                        # a whole block
    mask[10:30] = False                          # part of block 0 (crosses lane quads)
    mask[597] = False
    before = util.device_state(sh)
    o.gene_active = mask.copy()
    sh.set_gene_mask(mask)
    tr_o = o.minimize(P["counts_pc"], P["Xc"], 5, 0.01, 1)
    tr_d = sh.step(5, 0.01, 1)
    after = util.device_state(sh)
    np.testing.assert_allclose(tr_d, tr_o, rtol=3e-5)
    for k in ("Z_loc", "Z_std_log", "Wc_loc", "intercept", "sigma_log"):
        np.testing.assert_array_equal(after[k][:, ~mask], before[k][:, ~mask])          # frozen: bit-identical
        assert np.abs(after[k][:, mask] - before[k][:, mask]).max() > 0
    assert_states_close(util.oracle_state(o), after)
    win = sh.read_loss_window(6)
    np.testing.assert_array_equal(win[1:, ~mask], np.repeat(win[:1, ~mask], 5, axis=0))   # carried forward
    sh.set_gene_mask(None)
    o.gene_active[:] = True
    np.testing.assert_allclose(sh.step(2, 0.01, 1), o.minimize(P["counts_pc"], P["Xc"], 2, 0.01, 1), rtol=3e-5)
    assert_states_close(util.oracle_state(o), util.device_state(sh))


def test_packing_active_quads_is_bit_identical(lib, monkeypatch):
    """With frozen genes the active quads are packed to the front so that frozen 256-gene blocks are skipped
    (gather of every gene-indexed array, quad_ids keep the noise stream): every result must be bit-identical
    to leaving the genes in place (BRIE_PACK_ACTIVE=0)."""
    from brie_amd import _capi
    Nc, Ng, Kc = 50, 1100, 2                    # 5 gene blocks
    P = util.problem(Nc, Ng, Kc, 3, seed=79)
    rng = np.random.default_rng(3)
    mask = rng.random(Ng) < 0.3                 # 30 % of the genes stay active, scattered
    mask[900:] = False
    outs = []
    for pack in ("1", "0"):
        monkeypatch.setenv("BRIE_PACK_ACTIVE", pack)
        sh = util.device_shard(P, Nc, Ng, Kc, 83)
        sh.step(3, 0.01, 1)
        sh.set_gene_mask(mask)
        tr = sh.step(4, 0.01, 3)
        win = sh.read_loss_window(5)            # forces the identity order back
        tr2 = sh.step(2, 0.01, 1)               # still masked, packed again? (mask persists, order restored)
        sh.set_gene_mask(None)
        tr3 = sh.step(2, 0.01, 1)
        outs.append((util.device_state(sh), tr, win, tr2, tr3, sh.loss_gene(3), sh.read(_capi.PSI),
                     sh.read(_capi.COUNT3)))
        sh.close()
    a, b = outs
    for k in util.STATE_KEYS:
        np.testing.assert_array_equal(a[0][k], b[0][k])
    np.testing.assert_allclose(a[1], b[1], rtol=1e-6)          # fp64 block sums of the trace run in another order
    np.testing.assert_array_equal(a[2], b[2])
    np.testing.assert_allclose(a[3], b[3], rtol=1e-6)
    np.testing.assert_allclose(a[4], b[4], rtol=1e-6)
    np.testing.assert_array_equal(a[5], b[5])
    np.testing.assert_array_equal(a[6], b[6])
    np.testing.assert_array_equal(a[7], P["counts_pc"][2])


@pytest.mark.parametrize("Ng", [257, 258, 1027])
def test_packing_with_a_ragged_last_quad_keeps_every_frozen_loss(lib, monkeypatch, Ng):
    """Ng % 4 != 0: packing the active quads to the front moves a FULL quad into the last position, whose genes 1..3 then
    sit at positions >= Ng.  The per-gene finalize used to stop at position Ng and dropped the carried losses of those
    (frozen) genes from the trace and the loss ring -- found by soak sequence 61 of call r3c (profiles/history/r3d_soak_seq61_failure.log).
    Trace, loss window and state against the oracle, and bit-identical to the unpacked run."""
    Nc, Kc = 130, 3
    P = util.problem(Nc, Ng, Kc, 3, seed=361)
    rng = np.random.default_rng(961)
    mask = rng.random(Ng) < 0.5
    mask[Ng - 1] = True                                   # the ragged quad itself stays active and moves forward
    full = [q for q in range(Ng // 4) if not mask[4 * q:4 * q + 4].any()]
    assert full, "the mask must freeze at least one whole quad"
    outs = []
    for pack in ("1", "0"):
        monkeypatch.setenv("BRIE_PACK_ACTIVE", pack)
        o = util.oracle_model(P, Nc, Ng, Kc, 101, np.float32)
        sh = util.device_shard(P, Nc, Ng, Kc, 101)
        np.testing.assert_allclose(sh.step(2, 0.01, 2), o.minimize(P["counts_pc"], P["Xc"], 2, 0.01, 2), rtol=5e-5)
        o.gene_active = mask.copy()
        sh.set_gene_mask(mask)
        tr, tr_o = sh.step(3, 0.01, 2), o.minimize(P["counts_pc"], P["Xc"], 3, 0.01, 2)
        np.testing.assert_allclose(tr, tr_o, rtol=5e-5)
        win = sh.read_loss_window(4)
        np.testing.assert_allclose(win, np.asarray(o.lg_hist[-4:]), rtol=5e-5, atol=2e-3)
        assert (win[-1][~mask] != 0).all()                # every frozen gene still carries its last loss
        outs.append((tr, win, util.device_state(sh)))
        sh.close()
    np.testing.assert_allclose(outs[0][0], outs[1][0], rtol=1e-6)
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    for k in util.STATE_KEYS:
        np.testing.assert_array_equal(outs[0][2][k], outs[1][2][k])


def _random_cases(n, seed=20261001):
    """Seeded sweep over shapes and model switches (the same list on every run)."""
    rng = np.random.default_rng(seed)
    cases = []
    for i in range(n):
        Nc = int(rng.choice([1, 2, 7, 63, 64, 65, 130, 257, 300, 515]))
        Ng = int(rng.choice([1, 3, 4, 5, 255, 256, 257, 511, 700, 1025]))
        L = int(rng.choice([2, 3]))
        MC = int(rng.choice([1, 2, 3, 5]))
        kind = ["plain", "plain", "wide", "cell", "xg", "fixed", "margin", "wide_cell", "wide_xg"][i % 9]
        Kc = int(rng.integers(9, 41)) if kind.startswith("wide") else int(rng.integers(0, 9))
        Kg = int(rng.choice([1, 2, 3, 4, 5, 7, 12, 40])) if kind.endswith("xg") else 0
        eff = bool(L == 3 or rng.random() < 0.3)
        cases.append((i, kind, Nc, Ng, Kc, Kg, L, MC, eff))
    return cases


@pytest.mark.parametrize("i,kind,Nc,Ng,Kc,Kg,L,MC,eff", _random_cases(63))
def test_randomised_shapes_and_switches(lib, i, kind, Nc, Ng, Kc, Kg, L, MC, eff):
    """Every product switch of BRIE2.__init__/fit (model_TFProb.py:42-85,214-273) at awkward sizes: partly filled
    gene blocks, fewer cells than one chunk, single cell / single gene, Kc across the fused/wide boundary."""
    from brie_amd import _capi
    P = util.problem(Nc, Ng, Kc, L, seed=1000 + i)
    if not eff:
        P["effLen"] = None
    elif P["effLen"] is None:                       # two layers with effective lengths (model_TFProb.py:168-183)
        P["effLen"] = np.random.default_rng(i).uniform(50, 400, (Ng, 6)).astype(np.float32)
    if Kg:
        P["Xg"] = np.random.default_rng(i + 77).standard_normal((Ng, Kg)).astype(np.float32)
    mode = "cell" if kind in ("cell", "wide_cell") else "gene"
    fixed = dict(intercept=0.25, sigma=1.5) if kind == "fixed" else {}
    seed = 5000 + i
    o = util.oracle_model(P, Nc, Ng, Kc, seed, np.float32, Kg=Kg, mode=mode, **fixed)
    sh = util.device_shard(P, Nc, Ng, Kc, seed, Kg=Kg, mode=mode, **fixed)
    target = "marginLik" if kind == "margin" else "ELBO"
    sh.set_target(target)
    n = 4
    tr_o = o.minimize(P["counts_pc"], P["Xc"], n, 0.01, MC, target=target)
    tr_d = sh.step(n, 0.01, MC)
    np.testing.assert_allclose(tr_d, tr_o, rtol=5e-5, atol=1e-3)
    assert_states_close(util.oracle_state(o), util.device_state(sh))
    # derived arrays: the same rule -- everything within the bound except at most one sign-flipped element (a +-lr first
    # step of Z_loc / Z_std_log decided by a rounding error, see assert_states_close), itself bounded by what 2.2 lr can do
    # (Psi: sigmoid' <= 1/4; Z_std: relative; the CI width follows both)
    for which, want, bound, flip in ((_capi.PSI, o.Psi, 5e-4, 0.25 * 2.2 * 0.01), (_capi.PSI95CI, o.Psi95CI, 5e-4, 0.05)):
        d = np.abs(sh.read(which) - want)
        assert np.percentile(d, 99.9) < 2e-5 or d.size < 1000, (which, float(np.percentile(d, 99.9)))
        assert (d >= bound).sum() <= max(1, int(1e-4 * d.size)) and d.max() < flip, (which, int((d >= bound).sum()), float(d.max()))
    rel = np.abs(sh.read(_capi.Z_STD) - o.Z_std) / np.maximum(o.Z_std, 1e-6)
    assert (rel >= 1e-3).sum() <= max(1, int(1e-4 * rel.size)) and rel.max() < 2.5 * 0.01, (int((rel >= 1e-3).sum()), float(rel.max()))
    lg_o = o.eval_loss_gene(P["counts_pc"], P["Xc"], 3, target=target)          # 3-draw per-gene loss, updated state
    np.testing.assert_allclose(sh.loss_gene(3), lg_o, rtol=2e-4, atol=2e-3)
    sh.close()


def _fusable_cases():
    """The plain / fixed members of the random family (uncoupled, Kc <= 8, ELBO) + shapes of the kind brie-quant sees."""
    out = [(i, Nc, Ng, Kc, L, MC if MC in (1, 3) else (1 if i % 2 else 3), eff, kind == "fixed")
           for i, kind, Nc, Ng, Kc, Kg, L, MC, eff in _random_cases(63) if kind in ("plain", "fixed")]
    out += [(100, 200, 500, 0, 2, 1, False, False), (101, 200, 500, 1, 2, 3, False, False), (102, 300, 2000, 2, 3, 3, True, False),
            (103, 96, 1300, 8, 2, 1, True, False), (104, 1000, 300, 3, 3, 1, True, False)]
    return out


@pytest.mark.parametrize("i,Nc,Ng,Kc,L,MC,eff,fixed", _fusable_cases())
def test_many_steps_per_launch_are_bit_identical_to_the_two_launch_path(lib, i, Nc, Ng, Kc, L, MC, eff, fixed):
    """for small inputs brie_step runs its n steps as ONE launch (the PERSIST variant: barrier of the gene
    block per step, the per-gene Adam inside the kernel).  Against a second handle held to the two-launch path: loss traces,
    every state array, the loss ring and -- through further steps on both paths crosswise -- the Adam moments, bit for bit,
    over the stages of a staged fit (fresh optimiser per stage, model_TFProb.py:234-241), with a frozen-gene interlude that
    must fall back to the two-launch path by itself."""
    from brie_amd import _capi
    P = util.problem(Nc, Ng, Kc, L, seed=3000 + i)
    if not eff:
        P["effLen"] = None
    elif P["effLen"] is None:
        P["effLen"] = np.random.default_rng(i).uniform(50, 400, (Ng, 6)).astype(np.float32)
    kw = dict(intercept=0.25, sigma=1.5) if fixed else {}
    a = util.device_shard(P, Nc, Ng, Kc, 7000 + i, **kw)          # two launches per step
    b = util.device_shard(P, Nc, Ng, Kc, 7000 + i, **kw)          # one launch per brie_step
    a.set_step_fusion(0)
    b.set_step_fusion(1)

    def same(what):
        sa, sb = util.device_state(a), util.device_state(b)
        for k in util.STATE_KEYS:
            assert np.array_equal(sa[k], sb[k]), (what, k, util.max_abs_diff(sa[k], sb[k]))

    for n, lr in ((5, 0.001), (7, 0.02), (2, 0.005)):
        a.reset_optimizer(); b.reset_optimizer()
        ta, tb = a.step(n, lr, MC), b.step(n, lr, MC)
        assert np.array_equal(ta, tb), (n, lr, ta, tb)
        same("stage of %d steps" % n)
    info = b.step_fusion_info()
    # (a grid the device cannot hold at once -- beyond ~256 workgroups -- is declined by the launcher: two launches, as `a`)
    assert info in ({"launches": 3, "steps": 14}, {"launches": 0, "steps": 0}) and a.step_fusion_info()["launches"] == 0, info
    assert info["launches"] == 3 or Nc * Ng > 200 * 1000, (info, Nc, Ng)
    assert np.array_equal(a.read_loss_window(9), b.read_loss_window(9))                 # the loss ring of the last stages
    # the moments: continue WITHOUT a fresh optimiser, paths swapped
    a.set_step_fusion(1); b.set_step_fusion(0)
    assert np.array_equal(a.step(3, 0.01, MC), b.step(3, 0.01, MC))
    same("continued, paths swapped")
    np.testing.assert_array_equal(a.loss_gene(2), b.loss_gene(2))
    # frozen genes: the fused path steps aside (the carried losses of skipped blocks are the two-launch path's business)
    if Ng >= 8:
        mask = np.ones(Ng, np.uint8)
        mask[: Ng // 2] = 0
        a.set_gene_mask(mask); b.set_gene_mask(mask)
        before = a.step_fusion_info()["launches"]
        assert np.array_equal(a.step(2, 0.01, MC), b.step(2, 0.01, MC))
        assert a.step_fusion_info()["launches"] == before
        same("with frozen genes")
        a.set_gene_mask(None); b.set_gene_mask(None)
        assert np.array_equal(a.step(2, 0.01, MC), b.step(2, 0.01, MC))
        # (fused again unless packing the active quads left the counts in one u16 tier, which has no fused instantiation)
        assert a.step_fusion_info()["launches"] in (before, before + 1)
        same("mask cleared")
    a.close(); b.close()


def test_a_fused_launch_that_cannot_complete_ends_with_an_error_not_a_hang(lib):
    """The barrier wait of the many-steps-per-launch kernel is bounded: when a workgroup of a gene block never arrives (here:
    told not to, brie_debug_step_fusion, with a short poll bound) the kernel ends, and the next call on the handle fails loudly --
    the state is undefined -- instead of a kernel spinning on the GPU for ever."""
    import os
    import subprocess
    import sys
    code = ("import time\n"
            "t0 = time.time()\n"
            "import numpy as np\n"
            "from tests import util\n"
            "P = util.problem(200, 500, 0, 2, seed=5); P['effLen'] = None\n"
            "sh = util.device_shard(P, 200, 500, 0, 9)\n"
            "print('shard ready', round(time.time() - t0, 2), flush=True)\n"
            "sh.set_step_fusion(1)\n"
            "sh.debug_step_fusion(8 | (14 << 8))\n"
            "sh.step(4, 0.01, 1, trace=False)\n"
            "sh.synchronize()\n"
            "print('kernel ended', round(time.time() - t0, 2), flush=True)\n"
            "try:\n"
            "    sh.step(2, 0.01, 1)\n"
            "    print('NO ERROR')\n"
            "except RuntimeError as e:\n"
            "    print('ERROR:', e)\n"
            "print('done', round(time.time() - t0, 2), flush=True)\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    import time
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    elapsed = time.time() - t0
    if os.path.isdir(os.path.join(root, "gpurun_out")):
        with open(os.path.join(root, "gpurun_out", "cannot_complete_child.log"), "w") as f:
            f.write("elapsed %.1f s\n%s\n%s" % (elapsed, r.stdout, r.stderr[-2000:]))
    assert "ERROR:" in r.stdout and "gave up waiting" in r.stdout, r.stdout + r.stderr
    # (the poll bound is 2^14 here: the child needs ~2 s.  Two full-suite runs of round 6 -- calls r8y, r8z -- spent 141 s in this
    # test; three later runs with this log in place took 2 s each.  Candidate: the child's libraries paged in again after the
    # full-size tests' host buffers had pushed them out of the page cache.  The log is kept for the next time)


def test_fused_launches_of_several_handles_side_by_side(lib):
    """Six handles, a host thread and a stream each, their many-steps-per-launch kernels in flight TOGETHER: every one ends
    bit-identical to its two-launch twin.  (With at most 8 gene blocks a launch gives each gene block an XCD of its own; the
    handles start at different columns, so small fits running side by side do not all wait for the same XCD -- and a grid that
    waits for its workgroups behind another one must still complete.)"""
    import threading
    shapes = [(200, 500, 0, 2, 1), (200, 500, 1, 2, 3), (120, 900, 0, 2, 1), (64, 300, 2, 3, 3), (256, 2000, 0, 2, 1), (200, 1500, 3, 3, 3)]
    twins, fused = [], []
    for i, (Nc, Ng, Kc, L, MC) in enumerate(shapes):
        P = util.problem(Nc, Ng, Kc, L, seed=4100 + i)
        a = util.device_shard(P, Nc, Ng, Kc, 8100 + i)
        b = util.device_shard(P, Nc, Ng, Kc, 8100 + i)
        a.set_step_fusion(0)
        b.set_step_fusion(1)
        twins.append(a); fused.append(b)
    for a, (_, _, _, _, MC) in zip(twins, shapes):
        for _ in range(3):
            a.step(150, 0.01, MC, trace=False)
        a.synchronize()
    errors = []

    def run(b, MC):
        try:
            for _ in range(3):
                b.step(150, 0.01, MC, trace=False)
            b.synchronize()
        except Exception as e:                  # noqa: BLE001 -- reported by the asserting thread
            errors.append(repr(e))

    threads = [threading.Thread(target=run, args=(b, sh[4])) for b, sh in zip(fused, shapes)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for a, b, sh in zip(twins, fused, shapes):
        assert b.step_fusion_info()["launches"] == 3, (sh, b.step_fusion_info())
        sa, sb = util.device_state(a), util.device_state(b)
        for k in util.STATE_KEYS:
            assert np.array_equal(sa[k], sb[k]), (sh, k, util.max_abs_diff(sa[k], sb[k]))
        a.close(); b.close()


@pytest.mark.parametrize("Nc,Ng,Kc,L,MC,cuts", [
    (300, 1000, 3, 2, 1, (256, 512)),          # cuts on gene-block boundaries
    (515, 700, 2, 3, 3, (4, 260, 696)),        # cuts inside blocks, a 4-gene shard at each end
    (130, 1030, 12, 2, 2, (516,)),             # wide design (LDS tile + MFMA reduction)
    (64, 37, 0, 2, 1, (8, 36)),                # last shard = the ragged tail (1 gene)
])
def test_gene_shards_bit_identical_to_whole_fit(lib, Nc, Ng, Kc, L, MC, cuts):
    """genes are independent, so any contiguous split (boundaries multiples of 4 = one Philox quad)
    must reproduce the whole fit bit for bit -- state, per-gene loss and the summed loss trace (fp64 sum)."""
    P = util.problem(Nc, Ng, Kc, L, seed=77)
    whole = util.device_shard(P, Nc, Ng, Kc, 8)
    whole.step(6, 0.01, MC)
    lg = whole.loss_gene(4)
    a = util.device_state(whole)
    edges = (0,) + tuple(cuts) + (Ng,)
    for g0, g1 in zip(edges[:-1], edges[1:]):
        Ps = dict(P, counts=[c[:, g0:g1].copy() for c in P["counts"]],
                  effLen=None if P["effLen"] is None else P["effLen"][g0:g1].copy())
        part = util.device_shard(Ps, Nc, g1 - g0, Kc, 8, gene_offset=g0)
        part.step(6, 0.01, MC)
        b = util.device_state(part)
        for k in util.STATE_KEYS:
            if a[k].size:
                np.testing.assert_array_equal(a[k][:, g0:g1], b[k], err_msg="%s genes %d:%d" % (k, g0, g1))
        np.testing.assert_array_equal(lg[g0:g1], part.loss_gene(4))
        part.close()
    whole.close()


def _op_sequences(n, seed=777):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        Nc = int(rng.choice([5, 64, 130, 300]))
        Ng = int(rng.choice([4, 9, 257, 600, 1100]))
        Kc = int(rng.choice([0, 1, 3, 8, 11]))
        L = int(rng.choice([2, 3]))
        ops = [str(rng.choice(["step", "step", "mask", "unmask", "reset", "loss_gene", "tiling", "window", "read"]))
               for _ in range(8)]
        out.append((i, Nc, Ng, Kc, L, bool(rng.random() < 0.5), bool(rng.random() < 0.5), tuple(ops)))
    return out


@pytest.mark.parametrize("i,Nc,Ng,Kc,L,sparse,f32,ops", _op_sequences(40))
def test_randomised_operation_sequences(lib, i, Nc, Ng, Kc, L, sparse, f32, ops):
    """Random interleavings of the calls a fit makes (steps, per-batch masks incl. packing and un-packing, fresh
    optimiser, loss_gene, loss window, read-back, re-tiling) on random shapes, sparse or dense upload, compact or
    fp32 count storage -- the oracle follows with the same masks and the same noise-stream position."""
    import scipy.sparse as sp
    from brie_amd import _capi
    # Worst element: these sequences put "reset" (fresh Adam) in front of 1-2 step blocks -- the sign-flip exemption of
    # assert_states_close is sized by the number of fresh optimisers in the sequence.  The bulk bound (99.9 % within 2e-5) stays.
    fresh = 1 + sum(op == "reset" for op in ops)                  # every reset is a fresh optimiser (+ the initial one)
    rng = np.random.default_rng(900 + i)
    P = util.problem(Nc, Ng, Kc, L, seed=300 + i)
    if i % 2:                  # a few counts above 255 (moderate: a count of 60000 amplifies fp32 rounding past the state
                               # tolerances): count tiers per gene quad, u8 / u16 mixed
        P["counts"] = [c.copy() for c in P["counts"]]
        for _ in range(int(rng.integers(1, 6))):
            P["counts"][int(rng.integers(0, L))][int(rng.integers(0, Nc)), int(rng.integers(0, Ng))] = float(rng.integers(256, 3000))
        P["counts_pc"] = util.add_pseudo_count(P["counts"], 0.01)
    o = util.oracle_model(P, Nc, Ng, Kc, 40 + i, np.float32)
    sh = _capi.Shard(Nc, Ng, Kc, n_layers=L, has_efflen=P["effLen"] is not None, seed=40 + i)
    if f32:
        sh.set_count_storage(1)
    for l in range(L):
        sh.upload(_capi.COUNT1 + l, sp.csc_matrix(P["counts"][l]) if sparse else P["counts"][l])
    sh.add_pseudo_count(0.01)
    if P["effLen"] is not None:
        sh.upload(_capi.EFFLEN, P["effLen"])
    if Kc:
        sh.upload(_capi.XC, P["Xc"])
    sh.init_state()
    masked = False
    for op in ops:
        if op == "step":
            n, mc = int(rng.integers(1, 4)), int(rng.choice([1, 3, 2]))
            np.testing.assert_allclose(sh.step(n, 0.01, mc), o.minimize(P["counts_pc"], P["Xc"], n, 0.01, mc),
                                       rtol=5e-5, atol=1e-3, err_msg=str((op, ops)))
        elif op == "mask":
            if not o.lg_hist:
                continue                           # a mask needs a previous loss to carry forward
            mask = rng.random(Ng) < rng.choice([0.1, 0.5, 0.9])
            if Ng > 300:
                mask[256:300] = False
            o.gene_active = mask.copy()
            sh.set_gene_mask(mask)
            masked = True
        elif op == "unmask":
            o.gene_active = np.ones(Ng, bool)
            sh.set_gene_mask(None)
            masked = False
        elif op == "reset":
            o.reset_optimizer()
            sh.reset_optimizer()
        elif op == "loss_gene":
            np.testing.assert_allclose(sh.loss_gene(2), o.eval_loss_gene(P["counts_pc"], P["Xc"], 2),
                                       rtol=2e-4, atol=2e-3, err_msg=str((op, ops)))
        elif op == "tiling":
            sh.set_tiling(int(rng.choice([16, 32, 256])))
        elif op == "window" and o.lg_hist:
            k = min(len(o.lg_hist), 3)
            np.testing.assert_allclose(sh.read_loss_window(k), np.asarray(o.lg_hist[-k:]), rtol=5e-5, atol=2e-3)
        elif op == "read":
            assert_states_close(util.oracle_state(o), util.device_state(sh), fresh=fresh)
    assert_states_close(util.oracle_state(o), util.device_state(sh), fresh=fresh)
    np.testing.assert_array_equal(sh.read(_capi.COUNT1), P["counts_pc"][0])
    sh.close()


def test_distinct_handles_are_independent_across_threads(lib):
    """include/brie_amd.h: a handle is not thread-safe, distinct handles are.  Four host threads drive four shards
    (own HIP streams) at the same time; every result equals the same fit run alone."""
    import threading
    from brie_amd import _capi
    Nc, Ng, Kc = 70, 300, 2
    P = util.problem(Nc, Ng, Kc, 2, seed=91)

    def fit(seed, out, slot):
        sh = util.device_shard(P, Nc, Ng, Kc, seed)
        tr = [sh.step(5, 0.01, 1) for _ in range(6)]
        out[slot] = (np.concatenate(tr), sh.read(_capi.Z_LOC), sh.loss_gene(3))
        sh.close()

    alone, together = {}, {}
    for k in range(4):
        fit(100 + k, alone, k)
    threads = [threading.Thread(target=fit, args=(100 + k, together, k)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for k in range(4):
        for a, b in zip(alone[k], together[k]):
            np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("variant", ["cell_intercept", "gene_features_Kg2", "wide_Kc20_mfma_tile", "marginLik_mc3"])
def test_psi_after_full_default_schedule_model_variants(lib, variant):
    """The model variants beyond the plain per-gene model (gene features with per-cell weights, per-cell intercept /
    sigma, a wide cell design on the MFMA tile kernel, target="marginLik") through the WHOLE default schedule (6 x 166
    steps, fresh Adam per stage): HIP against the NumPy restatement in fp32, judged by the same comparison as everywhere
    else since round 4 -- what further fp32 CPU evaluations of the same algorithm do against that same oracle run.
    In the coupled models one sign event in a parameter shared by a row (a cell's Wg_loc entry, its intercept) reaches every
    gene of that cell, so two fp32 runs part across the whole matrix (21 % of the entries beyond 1e-4 with two gene
    features, round 3) and the gene-level partition of tests/util.py::psi_null_rule has no meaning; its ENTRY-level
    statistics and constants are applied to the whole matrix.  Which run draws the larger row / column events is a coin
    toss at 200 x 520, so the null is an ENSEMBLE, not a draw: five members of the family (OracleBRIE2 variant_b = 1 .. 5: float
    Box-Muller, reversed or blocked reductions, alone and combined) and the HIP figures are held against them by the ensemble
    rule of round 5 with its registered constants (tests/util.py::entry_ensemble_rule; the members' own leave-one-out record
    is printed beside the verdict)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from profiles.psi_delta_variants import run_variant, NULL_MEMBERS
    r = run_variant(variant, null=True)
    members = {"o32b%d" % v: r["o32b%d_vs_o32" % v] for v in NULL_MEMBERS}
    rep = util.entry_ensemble_rule(r["hip_vs_o32"], members, r["shape"][0] * r["shape"][1], variant, check=False)
    print(variant, rep)
    assert rep["holds"], (variant, rep)
