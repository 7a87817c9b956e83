"""-m gpu: the reference-shaped Python API (BRIE2 / fit_BRIE_matrix / fitBRIE) on the HIP path,
checked against the oracle run through the same host logic."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle.brie_oracle import OracleBRIE2, add_pseudo_count
from oracle.synth import make_problem
from tests.fakes import FakeAnnData, OracleBackedBRIE2

pytestmark = pytest.mark.gpu


def test_BRIE2_fit_matches_oracle_fit(lib):
    import brie_amd
    Nc, Ng, Kc = 120, 100, 1
    P = make_problem(Nc, Ng, Kc=Kc, L=2, seed=31, theta=2.0)
    m = brie_amd.BRIE2(Nc, Ng, Kc=Kc, seed=17)
    losses = m.fit(P["counts"], Xc=P["Xc"], min_iter=240, max_iter=240, n_loss_gene=20, pseudo_count=0.01,
                   verbose=False)
    o = OracleBRIE2(Nc, Ng, Kc, seed=17, dtype=np.float64)
    lo = o.fit(add_pseudo_count(P["counts"]), P["Xc"], min_iter=240, max_iter=240, n_loss_gene=20)
    assert losses.shape == lo.shape == (40,)                 # trace of the LAST stage only (model_TFProb.py:239)
    np.testing.assert_allclose(losses.numpy(), lo, rtol=2e-4)
    d = np.abs(m.Psi.numpy() - o.Psi)
    assert np.percentile(d, 99) < 1e-4 and d.max() < 2e-3
    np.testing.assert_allclose(m.loss_gene.numpy(), o.loss_gene, rtol=2e-3, atol=2e-2)
    np.testing.assert_allclose(m.Wc_loc.numpy(), o.Wc_loc, atol=2e-3)
    np.testing.assert_allclose(m.intercept.numpy(), o.intercept, atol=2e-3)
    assert m.Z_std.shape == (Nc, Ng) and m.Psi95CI.shape == (Nc, Ng) and m.sigma.shape == (1, Ng)
    assert isinstance(m.Psi95CI, np.ndarray) and m.n_iter == 240
    # get_loss (model_TFProb.py:194-211) on the fitted state: per gene and scalar, MC_size samples; both sides continue
    # the same noise stream, so the stochastic values agree too
    cnt = add_pseudo_count(P["counts"])
    lg = m.get_loss(P["counts"], axis=0, MC_size=4)
    np.testing.assert_allclose(lg.numpy(), o.eval_loss_gene(cnt, P["Xc"], 4), rtol=2e-3, atol=2e-2)
    tot = m.get_loss(P["counts"], MC_size=2)
    np.testing.assert_allclose(float(tot.numpy()), o.eval_loss_gene(cnt, P["Xc"], 2).sum(), rtol=1e-3)
    with pytest.raises(ValueError):
        m.get_loss(P["counts"], target="nonsense")
    # other data on the fitted model (the reference takes the layers per call, model_TFProb.py:194)
    other = [np.ascontiguousarray(c[::-1]) for c in P["counts"]]
    lg2 = m.get_loss(other, axis=0, MC_size=3)
    np.testing.assert_allclose(lg2.numpy(), o.eval_loss_gene(add_pseudo_count(other), P["Xc"], 3), rtol=2e-3, atol=2e-2)
    assert np.abs(lg2.numpy() - lg.numpy()).max() > 1.0
    # axis=1 (per cell) and target="marginLik" with several samples (model_TFProb.py:194-205: **kwargs reach logLik_MC,
    # any axis reaches reduce_sum): same evaluation, other reduction
    d0 = m._shard.draw
    g0 = m.get_loss(P["counts"], axis=0, MC_size=2)
    m._shard.draw = d0
    g1 = m.get_loss(P["counts"], axis=1, MC_size=2)
    assert g0.shape == (Ng,) and g1.shape == (Nc,)
    np.testing.assert_allclose(g1.numpy().astype(np.float64).sum(), g0.numpy().astype(np.float64).sum(), rtol=1e-4)
    m._shard.draw = d0
    o.draw = d0
    mg = m.get_loss(P["counts"], target="marginLik", axis=0, MC_size=3)
    np.testing.assert_allclose(mg.numpy(), o.margin_loss_and_grads(cnt, P["Xc"], 3, need_grads=False)["loss_gene"],
                               rtol=5e-3, atol=5e-2)        # (two separately fitted states: prior parameters within 2e-3)
    assert m._shard.draw == d0 + 1
    tot = m.get_loss(P["counts"], target="marginLik", MC_size=3)
    assert np.isfinite(float(tot.numpy())) and tot.numpy().shape == ()
    with pytest.raises(ValueError):
        m.get_loss(P["counts"], axis=2)
    m.close()


def test_BRIE2_init_obj_and_convergence_extension(lib):
    import brie_amd
    Nc, Ng = 60, 40
    P = make_problem(Nc, Ng, Kc=0, L=2, seed=5)
    rng = np.random.default_rng(1)
    init = dict(Z_loc=rng.standard_normal((Nc, Ng)).astype(np.float32),
                Z_std=np.exp(rng.standard_normal((Nc, Ng))).astype(np.float32),
                Wc_loc=np.zeros((0, Ng), np.float32), intercept=rng.standard_normal((1, Ng)).astype(np.float32),
                sigma=np.ones((1, Ng), np.float32))
    m = brie_amd.BRIE2(Nc, Ng, init_obj=init, seed=3)
    # epsilon below any loss drop => the while loop (model_TFProb.py:250-258) extends by add_iter until max_iter
    losses = m.fit(P["counts"], min_iter=120, max_iter=200, add_iter=20, epsilon_conv=-1e9, n_loss_gene=2,
                   verbose=False)
    assert m.n_iter == 200 and len(losses) == 20 + 80
    o = OracleBRIE2(Nc, Ng, 0, seed=3, dtype=np.float64,
                    init=dict(Z_loc=init["Z_loc"], Z_std_log=np.log(init["Z_std"]), Wc_loc=init["Wc_loc"],
                              intercept=init["intercept"], sigma_log=np.zeros((1, Ng))))
    lo = o.fit(P["counts"], None, min_iter=120, max_iter=200, add_iter=20, epsilon_conv=-1e9, n_loss_gene=2)
    assert o.n_iter == 200
    np.testing.assert_allclose(losses.numpy(), lo, rtol=2e-4)
    m.close()


def test_fit_BRIE_matrix_LRT_matches_oracle_backed_run(lib, monkeypatch):
    import brie_amd
    import brie_amd.models.wrap as wrap
    Nc, Ng, Kc = 150, 60, 2
    P = make_problem(Nc, Ng, Kc=Kc, L=3, seed=9, effect_frac=0.5, depth=6.0)
    kw = dict(Xc=P["Xc"], effLen=P["effLen"], LRT_index=[0], min_iter=180, max_iter=180, n_loss_gene=30,
              verbose=False, seed=4)
    res = brie_amd.fit_BRIE_matrix([sp.csc_matrix(c) for c in P["counts"]], **kw)     # sparse layers are densified
    monkeypatch.setattr(wrap, "BRIE2", OracleBackedBRIE2)
    ref = wrap.fit_BRIE_matrix([c.copy() for c in P["counts"]], **kw)
    np.testing.assert_allclose(res.ELBO_gain, ref.ELBO_gain, atol=0.05, rtol=5e-3)
    np.testing.assert_allclose(res.cell_coeff, ref.cell_coeff, atol=5e-3)
    d = np.abs(res.Psi - ref.Psi)
    assert np.percentile(d, 99) < 2e-4
    assert res.pval.shape == (Ng, 1) and res.fdr.shape == (Ng, 1)
    assert np.all((res.fdr >= res.pval - 1e-12) & (res.fdr <= 1))


def test_fitBRIE_end_to_end_recovers_effects(lib):
    """Data from the generative recipe (simulator.py:22-69): Psi recovered, LRT finds the effect genes."""
    import brie_amd
    Nc, Ng, Kc = 1500, 200, 1
    P = make_problem(Nc, Ng, Kc=Kc, L=2, seed=21, theta=1.0, depth=8.0, effect_frac=0.3)
    ad = FakeAnnData({'isoform1': P["counts"][0], 'isoform2': P["counts"][1]})
    res = brie_amd.fitBRIE(ad, Xc=P["Xc"], LRT_index=[0], min_iter=600, max_iter=600, n_loss_gene=100,
                           verbose=False, seed=2)
    assert np.corrcoef(ad.layers['Psi'].ravel(), P["Psi_true"].ravel())[0, 1] > 0.85
    strong = np.abs(P["W_true"][0]) > 0.7
    null = P["W_true"][0] == 0
    assert np.corrcoef(ad.varm['cell_coeff'][strong, 0], P["W_true"][0][strong])[0, 1] > 0.95
    gain = ad.varm['ELBO_gain'][:, 0]
    print("ELBO_gain median: strong %.2f null %.2f; fdr<0.05: strong %.2f null %.2f" % (
        np.median(gain[strong]), np.median(gain[null]), np.mean(ad.varm['fdr'][strong, 0] < 0.05),
        np.mean(ad.varm['fdr'][null, 0] < 0.05)))
    assert np.mean(ad.varm['fdr'][strong, 0] < 0.05) > 0.9          # power
    assert np.median(gain[strong]) > 10 * max(1.0, abs(np.median(gain[null])))
    assert ad.var['loss_gene'].shape == (Ng,) and res.losses.shape == (100,)
    assert set(ad.uns['brie_param']) >= {'LRT_index', 'base_mode', 'pseudo_count', 'layer_keys'}


def test_cli_quant_end_to_end_on_gpu(lib, tmp_path):
    """brie-quant front end -> fitBRIE -> HIP kernels -> result files (reference CLI: brie/bin/quant.py)."""
    import pandas as pd
    from brie_amd.cli.quant import main
    from tests.test_cli import _write_brie_npz
    Nc, Ng = 300, 64
    P = make_problem(Nc, Ng, Kc=1, L=3, seed=77, depth=10.0, effect_frac=0.4)
    in_file = str(tmp_path / "counts.npz")
    cells, genes = _write_brie_npz(in_file, P)
    cell_file = str(tmp_path / "cells.csv")
    with open(cell_file, "w") as f:
        f.write("cellID,group\n" + "".join("%s,%g\n" % (c, x) for c, x in zip(cells, P["Xc"][:, 0])))
    out = str(tmp_path / "out" / "brie_quant.h5ad")
    main(["-i", in_file, "-c", cell_file, "-o", out, "--LRTindex=All", "--interceptMode=gene", "--minCount=20",
          "--minUniqCount=5", "--minCell=10", "--minIter=300", "--maxIter=300", "--MCsize=3", "--seed=5"])
    df = pd.read_csv(str(tmp_path / "out" / "brie_quant.brie_ident.tsv"), sep="\t", index_col=0)
    assert {'n_counts', 'cdr', 'intercept', 'sigma', 'group_ceoff', 'group_ELBO_gain', 'group_pval', 'group_FDR'} <= set(df.columns)
    assert len(df) > 10 and np.all(np.isfinite(df['group_ELBO_gain'])) and df['intercept'].notna().all()
    bundle = np.load(str(tmp_path / "out" / "brie_quant.npz"), allow_pickle=True)
    psi = bundle["layers/Psi"]
    keep = [int(g[4:]) for g in df.index]
    assert psi.shape == (Nc, len(df)) and psi.min() > 0 and psi.max() < 1
    covered = (P["counts"][0] + P["counts"][1])[:, keep] > 8
    assert np.corrcoef(psi[covered], P["Psi_true"][:, keep][covered])[0, 1] > 0.7
    truth = P["W_true"][0][keep]
    strong = np.abs(truth) > 0.7
    if strong.sum() >= 3:
        assert np.corrcoef(df['group_ceoff'].values[strong], truth[strong])[0, 1] > 0.7


def test_fitBRIE_emulated_reference_batches_on_gpu(lib):
    """emulate_batches=True reproduces the reference's sequential gene batches (model_wrap.py:241-260): each
    batch has its own fit and loss trace; per-gene results equal the concurrent fit up to fp32 summation order."""
    import brie_amd
    Nc, Ng = 40, 36
    P = make_problem(Nc, Ng, Kc=1, L=2, seed=83)
    mk = lambda: FakeAnnData({'isoform1': P["counts"][0], 'isoform2': P["counts"][1]})
    kw = dict(Xc=P["Xc"], min_iter=120, max_iter=120, n_loss_gene=10, verbose=False, seed=3)
    whole = brie_amd.fitBRIE(mk(), **kw)
    batched = brie_amd.fitBRIE(mk(), batch_size=Nc * 12, emulate_batches=True, **kw)
    assert len(batched.losses) == 3 * len(whole.losses) and batched.Ng == Ng
    np.testing.assert_allclose(batched.Psi, whole.Psi, atol=1e-5)
    np.testing.assert_allclose(batched.loss_gene, whole.loss_gene, rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(batched.losses.reshape(3, -1).sum(0), whole.losses, rtol=1e-5)


def test_fitBRIE_per_batch_convergence_matches_oracle_backed_run(lib, monkeypatch):
    """Default fitBRIE: concurrent fit + per-batch stopping (model_wrap.py:241-260, model_TFProb.py:247-258)."""
    import brie_amd
    import brie_amd.models.wrap as wrap
    Nc, Ng = 60, 48
    P = make_problem(Nc, Ng, Kc=1, L=2, seed=14, depth=6.0)
    mk = lambda: FakeAnnData({'isoform1': P["counts"][0], 'isoform2': P["counts"][1]})
    kw = dict(Xc=P["Xc"], batch_size=Nc * 8, min_iter=120, max_iter=200, add_iter=10, epsilon_conv=0.1,
              n_loss_gene=5, verbose=False, seed=7)
    res = brie_amd.fitBRIE(mk(), **kw)
    monkeypatch.setattr(wrap, "BRIE2", OracleBackedBRIE2)
    OracleBackedBRIE2.instances = []
    ref = wrap.fitBRIE(mk(), **kw)
    n_ref = OracleBackedBRIE2.instances[0]._o.n_iter_batch
    print("per-batch n_iter (oracle):", n_ref, "trace lengths:", len(res.losses), len(ref.losses))
    # the stop/continue decisions sit on fp32 loss differences: allow one extension of slack overall
    assert abs(len(res.losses) - len(ref.losses)) <= 10
    if len(res.losses) == len(ref.losses):
        d = np.abs(res.Psi - ref.Psi)
        assert np.percentile(d, 99) < 2e-4


def test_fitBRIE_super_batches_on_gpu(lib):
    """A gene range larger than the device is fitted as sequential super-batches (whole gene blocks); per-gene
    results equal the unsplit fit bit for bit, and 'auto' leaves a problem that fits alone."""
    import brie_amd
    import brie_amd.models.wrap as wrap
    Nc, Ng = 40, 700
    P = make_problem(Nc, Ng, Kc=1, L=2, seed=14)
    mk = lambda: FakeAnnData({'isoform1': P["counts"][0].copy(), 'isoform2': P["counts"][1].copy()})
    kw = dict(Xc=P["Xc"], LRT_index=[0], min_iter=120, max_iter=120, n_loss_gene=4, seed=5, verbose=False)
    whole = brie_amd.fitBRIE(mk(), **kw)                               # max_genes_per_fit='auto': fits
    ad = mk()
    split = brie_amd.fitBRIE(ad, max_genes_per_fit=300, **kw)          # -> 256-gene super-batches: 256 + 256 + 188
    for key in ("Psi", "Psi95CI", "Z_std", "cell_coeff", "sigma", "intercept", "loss_gene", "ELBO_gain", "pval", "fdr"):
        np.testing.assert_array_equal(getattr(split, key), getattr(whole, key), err_msg=key)
    assert ad.layers['Psi'].shape == (Nc, Ng) and len(split.losses) == 3 * len(whole.losses)
    free, total = brie_amd._capi.device_memory(0)
    assert 0 < free <= total and total > (100 << 30)
    assert wrap._super_batch_genes('auto', Nc, Ng, 2, 1, 0, 12500) is None


def test_common_noise_reduces_the_monte_carlo_error_of_ELBO_gain(lib):
    """fit_BRIE_matrix(common_noise=True): base and test model share seed and loss_gene draws, so for genes WITHOUT an
    effect the ELBO gain (model_wrap.py:155-187) is no longer dominated by the independent Monte-Carlo errors of two
    500-draw (here 40-draw) averages."""
    import brie_amd
    Nc, Ng = 1500, 64
    rng = np.random.default_rng(5)
    Xc = np.stack([(rng.random(Nc) < 0.5).astype(np.float32), rng.standard_normal(Nc).astype(np.float32)], axis=1)
    z = rng.normal(0, 1.5, (1, Ng)) + rng.normal(0, 1.0, (Nc, Ng))                 # no dependence on Xc at all
    psi = 1 / (1 + np.exp(-z))
    depth = rng.poisson(6.0, (Nc, Ng))
    c1 = rng.binomial(depth, psi).astype(np.float32)
    data = [c1, (depth - c1).astype(np.float32)]
    kw = dict(Xc=Xc, LRT_index=[0], min_iter=1500, max_iter=1500, n_loss_gene=40, seed=2, verbose=False)
    indep = brie_amd.fit_BRIE_matrix(data, **kw)
    common = brie_amd.fit_BRIE_matrix(data, common_noise=True, **kw)
    s_i, s_c = np.std(indep.ELBO_gain[:, 0]), np.std(common.ELBO_gain[:, 0])
    print("std of ELBO_gain on null genes: independent %.3f, common noise %.3f" % (s_i, s_c))
    assert s_c < 0.5 * s_i
    np.testing.assert_allclose(common.cell_coeff, indep.cell_coeff, atol=0.2)       # same model, same estimates


def test_n_iter_schedule_repeats_an_earlier_fit(lib):
    import brie_amd
    Nc, Ng = 80, 48
    P = make_problem(Nc, Ng, Kc=1, L=2, seed=23)
    kw = dict(Xc=P["Xc"], min_iter=300, max_iter=700, add_iter=40, epsilon_conv=0.3, n_loss_gene=3,
              pseudo_count=0.01, verbose=False, conv_batch_genes=8)
    a = brie_amd.BRIE2(Nc, Ng, Kc=1, seed=1)
    a.fit(P["counts"], **kw)
    assert len(set(a.n_iter_batch)) > 1                                   # batches stopped at different times
    b = brie_amd.BRIE2(Nc, Ng, Kc=1, seed=9)                              # other noise: own decisions would differ
    b.fit(P["counts"], n_iter_schedule=a.n_iter_batch, **dict(kw, epsilon_conv=-1e9))
    np.testing.assert_array_equal(b.n_iter_batch, a.n_iter_batch)
    c = brie_amd.BRIE2(Nc, Ng, Kc=1, seed=9)                              # global rule: run exactly as long as `a`
    c.fit(P["counts"], n_iter_schedule=[a.n_iter], **dict(kw, conv_batch_genes=None, epsilon_conv=1e9))
    assert c.n_iter == a.n_iter and c.n_iter_batch is None
    for m in (a, b, c):
        m.close()


def test_count_layers_in_any_container_give_the_same_fit(lib):
    """model_wrap.py:108-111 accepts whatever `.toarray()` / np.asarray understand; so does the upload path: integer
    and float64 arrays, Fortran order, scipy COO / CSR, torch tensors on the host -- all bit-identical fits."""
    import torch
    import brie_amd
    Nc, Ng = 50, 90
    P = make_problem(Nc, Ng, Kc=1, L=2, seed=6)
    variants = {
        "float32": lambda c: c,
        "int32": lambda c: c.astype(np.int32),               # integer / float64 layers: brie_upload_typed, no numpy cast
        "int64": lambda c: c.astype(np.int64),
        "uint16": lambda c: c.astype(np.uint16),
        "float64": lambda c: c.astype(np.float64),
        "int32_row_view": lambda c: np.concatenate([c, c], axis=1).astype(np.int32)[:, :Ng],   # row pitch > row length
        "float64_fortran": lambda c: np.asfortranarray(c.astype(np.float64)),
        "coo": lambda c: sp.coo_matrix(c),
        "csr": lambda c: sp.csr_matrix(c),
        "torch_cpu": lambda c: torch.from_numpy(c.copy()),
        "strided_view": lambda c: np.concatenate([c, c], axis=1)[:, :Ng],
    }
    ref = None
    for name, conv in variants.items():
        m = brie_amd.BRIE2(Nc, Ng, Kc=1, seed=4)
        m.fit([conv(c) for c in P["counts"]], Xc=P["Xc"], min_iter=60, max_iter=60, n_loss_gene=2, pseudo_count=0.01,
              verbose=False)
        got = (m.Psi.numpy(), m.loss_gene.numpy(), m.losses.numpy())
        m.close()
        if ref is None:
            ref = got
        for a, b in zip(ref, got):
            np.testing.assert_array_equal(a, b, err_msg=name)


@pytest.mark.parametrize("egress", ["two_streams", "one_stream"])
@pytest.mark.parametrize("Ng", [300, 1028, 1031])
def test_async_result_export_equals_the_plain_reads(lib, Ng, egress, monkeypatch):
    """brie_read_results_async: one pass over the state, slab by slab next to the main stream, overlapping loss_gene;
    the four matrices equal brie_read's, a state-changing call waits for the pending export first.
    "two_streams" (default): slab k on stream k & 1, its export kernel enqueued before the copies of slab k - 1 start;
    "one_stream": round 2's order (BRIE_IO_ONE_STREAM, A/B runs)."""
    from brie_amd import _capi
    from tests import util
    Nc, Kc = 70, 1
    P = util.problem(Nc, Ng, Kc, 2, seed=5)
    sh = util.device_shard(P, Nc, Ng, Kc, 3)
    sh.step(7, 0.01, 1)
    twin = util.device_shard(P, Nc, Ng, Kc, 3)                         # the same fit without any asynchronous export
    twin.step(7, 0.01, 1)
    monkeypatch.setenv("BRIE_IO_SLAB_ELEMS", str(Ng * 16))            # 16-row slabs: 5 slabs, the last one ragged
    if egress == "one_stream":
        monkeypatch.setenv("BRIE_IO_ONE_STREAM", "1")
    want = {w: sh.read(w) for w in (_capi.PSI, _capi.Z_STD, _capi.PSI95CI, _capi.Z_LOC)}
    got = {w: np.full((Nc, Ng), np.nan, np.float32) for w in want}
    _capi.host_register(got[_capi.PSI])                                # one page-locked destination, three pageable
    sh.read_results_async(got[_capi.PSI], got[_capi.Z_STD], got[_capi.PSI95CI], got[_capi.Z_LOC])
    lg = sh.loss_gene(3)                                               # runs NEXT TO the export (it only reads the state)
    np.testing.assert_array_equal(lg, twin.loss_gene(3))
    sh.step(2, 0.01, 1)                                                # must not run before the export has finished
    sh.read_wait()
    _capi.host_unregister(got[_capi.PSI])
    for w in want:
        np.testing.assert_array_equal(got[w], want[w])
    assert np.all(np.isfinite(lg)) and not np.array_equal(sh.read(_capi.PSI), want[_capi.PSI])
    # a packed gene order (per-batch convergence) has to be undone before loss_gene: that moves the state, so the
    # pending export is waited for first -- results as without it
    twin.step(2, 0.01, 1)
    mask = np.zeros(Ng, bool)
    mask[: max(4, Ng // 8)] = True
    for s_ in (sh, twin):
        s_.set_gene_mask(mask)
        s_.step(2, 0.01, 1)
    want2 = {w: twin.read(w) for w in want}
    sh.read_results_async(got[_capi.PSI], got[_capi.Z_STD], got[_capi.PSI95CI], got[_capi.Z_LOC])
    np.testing.assert_array_equal(sh.loss_gene(2), twin.loss_gene(2))
    sh.read_wait()
    for w in want:
        np.testing.assert_array_equal(got[w], want2[w])
    for s_ in (sh, twin):
        s_.set_gene_mask(None)
    twin.close()
    only = np.empty((Nc, Ng), np.float32)
    sh.read_results_async(psi95ci=only)                                # any subset
    sh.read_wait()
    np.testing.assert_array_equal(only, sh.read(_capi.PSI95CI))
    with pytest.raises(ValueError):
        sh.read_results_async(psi=np.empty((Nc, Ng + 1), np.float32))
    sh.close()


def test_successor_shard_reuses_the_device_arrays_and_starts_clean(lib):
    """brie_destroy keeps the cell x gene arrays (>= 256 MB each) for the next handle of the same size
    (include/brie_amd.h, brie_trim_memory): the successor must start from zeroed arrays -- same results as the first --,
    a handle of another size or brie_device_memory must give the memory back."""
    from brie_amd import _capi
    from tests import util
    Nc, Ng, Kc = 70000, 1000, 1                       # 70000 x 1024 x 4 B = 287 MB per array: above the cache's minimum
    rng = np.random.default_rng(77)                   # (plain small integers: the generative recipe takes 30 s at this size)
    P = {"counts": [rng.integers(0, 4, (Nc, Ng), dtype=np.uint8).astype(np.float32) for _ in range(2)], "effLen": None,
         "Xc": rng.standard_normal((Nc, Kc)).astype(np.float32)}
    _capi.trim_memory()
    free0 = _capi.device_memory()[0]

    def run():
        sh = util.device_shard(P, Nc, Ng, Kc, 9)
        tr = sh.step(4, 0.01, 1)
        out = (tr, sh.read(_capi.Z_LOC), sh.loss_gene(2))
        sh.close()
        return out
    a = run()
    b = run()                                         # on the arrays the first one left behind
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
    # the query releases what was kept (9 arrays of 287 MB); what remains are the process's staging lanes and streams
    assert abs(_capi.device_memory()[0] - free0) < (768 << 20)
    small = util.device_shard(util.problem(64, 72, 1, 2), 64, 72, 1, 3)           # another size: nothing of the old generation survives
    small.step(1, 0.01, 1)
    small.close()
    c = run()
    np.testing.assert_array_equal(a[1], c[1])
    _capi.trim_memory()


def test_BRIE2_fit_streams_results_out(lib):
    import brie_amd
    from brie_amd import _capi
    Nc, Ng = 80, 200
    P = make_problem(Nc, Ng, Kc=1, L=2, seed=3)
    m = brie_amd.BRIE2(Nc, Ng, Kc=1, seed=9)
    m.fit(P["counts"], Xc=P["Xc"], min_iter=60, max_iter=60, n_loss_gene=5, pseudo_count=0.01, verbose=False)
    assert m._results is not None
    for attr, which in (("Psi", _capi.PSI), ("Z_std", _capi.Z_STD), ("Psi95CI", _capi.PSI95CI), ("Z_loc", _capi.Z_LOC)):
        np.testing.assert_array_equal(np.asarray(getattr(m, attr)), m._shard.read(which))
    m2 = brie_amd.BRIE2(Nc, Ng, Kc=1, seed=9)
    m2.fit(P["counts"], Xc=P["Xc"], min_iter=60, max_iter=60, n_loss_gene=5, pseudo_count=0.01, verbose=False,
           prefetch_results=False)
    assert m2._results is None
    np.testing.assert_array_equal(np.asarray(m2.Psi), np.asarray(m.Psi))
    np.testing.assert_array_equal(m2.loss_gene, m.loss_gene)
    m.close(); m2.close()


def test_lrt_reuses_the_count_layers_bit_identically(lib, monkeypatch):
    """LRT with the handle handed from model to model (brie_reconfigure; counts uploaded / compacted once) against
    the same test with a fresh handle per model: every output bit for bit; also across a change of kernel variant
    (Kc 9 -> 8: wide LDS path to register path)."""
    import brie_amd
    import brie_amd.models.wrap as wrap
    from brie_amd.models.engine import BRIE2

    class Fresh(BRIE2):                       # ignores `reuse`: one new handle per model, as in round 1
        def __init__(self, *a, **kw):
            kw.pop("reuse", None)
            BRIE2.__init__(self, *a, **kw)
    for Kc, L in ((2, 3), (9, 2)):
        Nc, Ng = 120, 300
        P = make_problem(Nc, Ng, Kc=Kc, L=L, seed=9, effect_frac=0.5, depth=6.0)
        kw = dict(Xc=P["Xc"], effLen=P["effLen"], LRT_index=[0, Kc - 1], min_iter=120, max_iter=120, n_loss_gene=10,
                  verbose=False, seed=4)
        res = brie_amd.fit_BRIE_matrix([sp.csc_matrix(c) for c in P["counts"]], **kw)
        monkeypatch.setattr(wrap, "BRIE2", Fresh)
        ref = wrap.fit_BRIE_matrix([sp.csc_matrix(c) for c in P["counts"]], **kw)
        monkeypatch.setattr(wrap, "BRIE2", BRIE2)
        for key in ("ELBO_gain", "pval", "fdr", "Psi", "cell_coeff", "loss_gene", "sigma"):
            np.testing.assert_array_equal(getattr(res, key), getattr(ref, key), err_msg="Kc=%d %s" % (Kc, key))
