"""CPU: C-ABI surface, host statistics, gene sharding, and the fit_BRIE_matrix / fitBRIE
orchestration (with the oracle-backed stand-in from tests/fakes.py)."""
import os
import re

import numpy as np
import pytest

from oracle import host_stats
from oracle.synth import make_problem
from tests.fakes import FakeAnnData, OracleBackedBRIE2

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    from brie_amd.build import compile_library
    from brie_amd import _capi
    compile_library()                     # hipcc cross-compiles gfx950 without a GPU
    return _capi.load_library()


def test_capi_exports_every_declared_symbol(built_lib):
    from brie_amd import _capi
    header = open(os.path.join(ROOT, "include", "brie_amd.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = sorted(set(re.findall(r"\b(brie_[a-z_0-9]+)\s*\(", header)))
    assert declared, "no prototypes parsed"
    for name in declared:
        assert hasattr(built_lib, name), "libbrie_amd.so does not export %s" % name
    assert sorted(_capi.EXPORTS) == declared
    assert built_lib.brie_abi_version() == _capi.ABI_VERSION
    assert built_lib.brie_last_error() is not None


def test_capi_struct_layout_matches_header():
    import ctypes
    from brie_amd import _capi
    assert _capi.MAX_KC == 1024 and _capi.MAX_KG == 1024     # BRIE_MAX_KC_PANELS, BRIE_MAX_KG_PANELS of include/brie_amd.h
    assert ctypes.sizeof(_capi.BrieProblem) == 72
    assert _capi.BrieProblem.seed.offset == 64 and _capi.BrieProblem.Kc.offset == 32


def test_missing_library_fails_loudly(tmp_path):
    from brie_amd import _capi
    with pytest.raises(ImportError):
        _capi.load_library(str(tmp_path / "nope.so"))


def test_fdr_bh_matches_restatement_and_scipy():
    from scipy.stats import false_discovery_control
    from brie_amd.stats import fdr_bh, elbo_gain_pval
    rng = np.random.default_rng(0)
    for n in (1, 2, 17, 500):
        p = rng.uniform(size=n) ** 3
        p[rng.integers(0, n)] = p[0]                              # a tie
        np.testing.assert_allclose(fdr_bh(p), host_stats.fdr_bh(p), rtol=1e-12)
        np.testing.assert_allclose(fdr_bh(p), false_discovery_control(p, method="bh"), rtol=1e-12)
    gain = rng.normal(size=(30, 2)) * 3
    np.testing.assert_allclose(elbo_gain_pval(gain), host_stats.pval_from_gain(gain))
    assert elbo_gain_pval(np.array([0.0]))[0] == 1.0


def test_gene_shard_partition():
    from brie_amd.sharding import gene_shard
    for Ng, world in [(20000, 8), (20000, 1), (30000, 8), (500, 3), (7, 4), (5, 8)]:
        edges = [gene_shard(Ng, r, world) for r in range(world)]
        assert edges[0][0] == 0 and edges[-1][1] == Ng
        for (a0, a1), (b0, b1) in zip(edges, edges[1:]):
            assert a1 == b0 and a0 % 4 == 0 and b0 % 4 == 0 or b0 == Ng
        assert sum(b - a for a, b in edges) == Ng
    assert gene_shard(20000, 3, 8) == (7500, 10000)


@pytest.fixture
def patched_wrap(monkeypatch):
    import brie_amd.models.wrap as wrap
    OracleBackedBRIE2.instances = []
    monkeypatch.setattr(wrap, "BRIE2", OracleBackedBRIE2)
    return wrap


FIT = dict(min_iter=60, max_iter=60, n_loss_gene=3, verbose=False)


def test_fit_brie_matrix_lrt_bookkeeping_full_mode(patched_wrap):
    Nc, Ng, Kc = 40, 12, 2
    P = make_problem(Nc, Ng, Kc=Kc, L=2, seed=3)
    raw = [c.copy() for c in P["counts"]]
    res = patched_wrap.fit_BRIE_matrix(P["counts"], Xc=P["Xc"], LRT_index=[0, 1], **FIT)
    base, t0, t1 = OracleBackedBRIE2.instances
    assert base.fit_args["Kc"] == 2 and t0.fit_args["Kc"] == 1 and t1.fit_args["Kc"] == 1
    np.testing.assert_array_equal(t0.fit_args["Xc"], P["Xc"][:, 1:2])      # feature 0 deleted
    np.testing.assert_array_equal(t1.fit_args["Xc"], P["Xc"][:, 0:1])
    np.testing.assert_allclose(res.ELBO_gain[:, 0], t0.loss_gene - base.loss_gene, rtol=1e-6)
    np.testing.assert_allclose(res.pval, host_stats.pval_from_gain(res.ELBO_gain))
    for i in range(2):
        np.testing.assert_allclose(res.fdr[:, i], host_stats.fdr_bh(res.pval[:, i]))
    assert res.cell_coeff.shape == (2, Ng) and res.Psi.shape == (Nc, Ng)
    for a, b in zip(raw, P["counts"]):
        np.testing.assert_array_equal(a, b)                                 # caller arrays untouched


def test_fit_brie_matrix_null_mode_and_defaults(patched_wrap):
    Nc, Ng, Kc = 30, 8, 2
    P = make_problem(Nc, Ng, Kc=Kc, L=2, seed=4)
    res = patched_wrap.fit_BRIE_matrix(P["counts"], Xc=P["Xc"], LRT_index=[1], base_mode='null', **FIT)
    base, test = OracleBackedBRIE2.instances
    assert base.fit_args["Kc"] == 1 and test.fit_args["Kc"] == 2
    np.testing.assert_array_equal(test.fit_args["Xc"][:, -1], P["Xc"][:, 1])
    np.testing.assert_allclose(res.ELBO_gain[:, 0], base.loss_gene - test.loss_gene, rtol=1e-6)
    assert res.cell_coeff.shape == (2, Ng)                                  # test weight appended
    OracleBackedBRIE2.instances = []
    res = patched_wrap.fit_BRIE_matrix(P["counts"], Xc=P["Xc"], LRT_index=None, **FIT)
    assert len(OracleBackedBRIE2.instances) == 3 and res.ELBO_gain.shape == (Ng, 2)   # None => all features
    OracleBackedBRIE2.instances = []
    res = patched_wrap.fit_BRIE_matrix(P["counts"], Xc=P["Xc"], LRT_index=[], **FIT)
    assert len(OracleBackedBRIE2.instances) == 1 and not hasattr(res, "ELBO_gain")


def test_fitBRIE_writes_the_reference_keys(patched_wrap):
    Nc, Ng, Kc = 36, 20, 1
    P = make_problem(Nc, Ng, Kc=Kc, L=3, seed=6)
    ad = FakeAnnData({'isoform1': P["counts"][0], 'isoform2': P["counts"][1], 'ambiguous': P["counts"][2]},
                     effLen=P["effLen"])
    res = patched_wrap.fitBRIE(ad, Xc=P["Xc"], LRT_index=[0], **FIT)
    assert set(ad.layers) >= {'Psi', 'Z_std', 'Psi_95CI'}
    assert ad.layers['Psi'].shape == (Nc, Ng)
    assert ad.varm['cell_coeff'].shape == (Ng, Kc) and ad.varm['intercept'].shape == (Ng, 1)
    assert ad.varm['sigma'].shape == (Ng, 1) and ad.var['loss_gene'].shape == (Ng,)
    assert ad.varm['ELBO_gain'].shape == (Ng, 1) and ad.varm['pval'].shape == (Ng, 1)
    assert ad.varm['fdr'].shape == (Ng, 1)
    assert ad.uns['brie_param']['pseudo_count'] == 0.01 and 'brie_losses' in ad.uns
    assert res.Ng == Ng and str(res) == "BRIE2 results for %d cells and %d genes" % (Nc, Ng)
    # no LRT => no fdr keys
    ad2 = FakeAnnData({'isoform1': P["counts"][0], 'isoform2': P["counts"][1]})
    patched_wrap.fitBRIE(ad2, Xc=P["Xc"], **FIT)
    assert 'fdr' not in ad2.varm and 'Psi' in ad2.layers


def test_fitBRIE_emulated_batches_equal_whole_fit(patched_wrap):
    """Genes are independent (model_wrap.py:241): reference-style sequential gene batches give
    the same per-gene results as one concurrent fit, because the noise stream is keyed by the
    global gene index."""
    Nc, Ng = 25, 24
    P = make_problem(Nc, Ng, Kc=1, L=2, seed=8)
    mk = lambda: FakeAnnData({'isoform1': P["counts"][0], 'isoform2': P["counts"][1]})
    whole = patched_wrap.fitBRIE(mk(), Xc=P["Xc"], **FIT)
    batched = patched_wrap.fitBRIE(mk(), Xc=P["Xc"], batch_size=Nc * 8, emulate_batches=True, **FIT)
    assert len(OracleBackedBRIE2.instances) == 1 + 3
    np.testing.assert_allclose(batched.Psi, whole.Psi, atol=2e-6)
    np.testing.assert_allclose(batched.cell_coeff, whole.cell_coeff, atol=2e-6)
    assert len(batched.losses) == 3 * len(whole.losses)                     # traces concatenated (model_wrap.py:61)


def test_super_batch_sizing():
    """Ranges larger than the device are fitted as sequential super-batches: whole gene blocks (256) and whole
    convergence batches, as many as fit into 90 % of the free HBM."""
    import brie_amd.models.wrap as wrap
    from brie_amd import _capi

    class Mem(object):
        free = 0

        @staticmethod
        def free_device_memory(device=0):
            return Mem.free
    saved = wrap.BRIE2
    wrap.BRIE2 = Mem
    try:
        Nc, Ng = 50000, 20000
        Mem.free = 288 << 30
        assert wrap._super_batch_genes('auto', Nc, Ng, 2, 3, 0, 10) is None              # C3 fits an MI355X whole
        Mem.free = 24 << 30                                                              # a 24 GB card would not
        sb = wrap._super_batch_genes('auto', Nc, Ng, 2, 3, 0, 10)
        assert sb % 1280 == 0 and 0 < sb < Ng                                            # lcm(256, 10)
        assert _capi.shard_bytes(Nc, sb, 2, 3) <= 0.9 * Mem.free < _capi.shard_bytes(Nc, sb + 1280, 2, 3)
        assert wrap._super_batch_genes(None, Nc, Ng, 2, 3, 0, 10) is None
        assert wrap._super_batch_genes(5000, Nc, Ng, 2, 3, 0, 10) == 3840                # explicit limit, rounded down
        assert wrap._super_batch_genes(100, Nc, Ng, 2, 3, 0, 10) == 100                  # below one unit: multiples of 4
        assert wrap._super_batch_genes(50000, Nc, Ng, 2, 3, 0, 10) is None
        with pytest.raises(ValueError):
            wrap._super_batch_genes('sometimes', Nc, Ng, 2, 3, 0, 10)
    finally:
        wrap.BRIE2 = saved


def test_fitBRIE_super_batches_equal_whole_fit(patched_wrap):
    Nc, Ng = 25, 24
    P = make_problem(Nc, Ng, Kc=1, L=2, seed=8)
    mk = lambda: FakeAnnData({'isoform1': P["counts"][0], 'isoform2': P["counts"][1]})
    whole = patched_wrap.fitBRIE(mk(), Xc=P["Xc"], LRT_index=[0], **FIT)
    n0 = len(OracleBackedBRIE2.instances)
    split = patched_wrap.fitBRIE(mk(), Xc=P["Xc"], LRT_index=[0], max_genes_per_fit=8, **FIT)
    assert len(OracleBackedBRIE2.instances) - n0 == 3 * 2                               # 3 super-batches x (base + test)
    np.testing.assert_allclose(split.Psi, whole.Psi, atol=2e-6)
    np.testing.assert_allclose(split.ELBO_gain, whole.ELBO_gain, rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(split.fdr, whole.fdr, rtol=1e-3, atol=1e-6)           # BH over all genes, not per part


def test_more_ranks_than_gene_quads_is_refused(patched_wrap):
    class Comm(object):
        rank, world = 0, 8
    P = make_problem(10, 8, Kc=0, L=2, seed=1)
    ad = FakeAnnData({'isoform1': P["counts"][0], 'isoform2': P["counts"][1]})
    with pytest.raises(ValueError, match="cannot be sharded"):
        patched_wrap.fitBRIE(ad, comm=Comm(), **FIT)


@pytest.mark.parametrize("coupling", ["Kg", "cell"])
@pytest.mark.parametrize("emulate", [False, True])
def test_coupled_sharded_fit_always_decides_on_the_global_loss(patched_wrap, monkeypatch, coupling, emulate):
    """a COUPLED model (Kg > 0 or intercept_mode='cell') ignores emulate_batches -- it stays one sharded fit
    with a per-step all-reduce -- so its stopping decisions must be taken on the loss summed over ranks whatever
    emulate_batches says; a rank-local decision lets the ranks extend a different number of rounds and their
    collectives no longer pair up.  Only the literal independent batch loop (separable fits) drops the collectives."""
    class Comm(object):
        rank, world = 0, 2

        def allreduce_sum(self, a):
            return np.asarray(a, np.float64)
        allreduce_min = allreduce_sum

    class Stop(Exception):
        pass
    seen = {}

    def recorder(data, **kw):
        seen.update(kw)
        raise Stop()
    monkeypatch.setattr(patched_wrap, "fit_BRIE_matrix", recorder)
    P = make_problem(12, 16, Kc=1, L=2, seed=3)
    ad = FakeAnnData({'isoform1': P["counts"][0], 'isoform2': P["counts"][1]})
    kw = dict(Xg=np.ones((16, 1), np.float32)) if coupling == "Kg" else dict(intercept_mode='cell')
    comm = Comm()
    with pytest.raises(Stop):
        patched_wrap.fitBRIE(ad, Xc=P["Xc"], comm=comm, emulate_batches=emulate, **dict(FIT, **kw))
    assert seen["trace_reduce"].__func__ is Comm.allreduce_sum and seen["conv_total_genes"] == 16
    assert seen["comm"] is comm                                   # the per-step all-reduce of the coupled shard
    # the separable twin: literal batches exchange nothing, the concurrent fit does
    seen.clear()
    with pytest.raises(Stop):
        patched_wrap.fitBRIE(ad, Xc=P["Xc"], comm=comm, emulate_batches=emulate, **FIT)
    assert ("trace_reduce" in seen) == (not emulate) and seen["comm"] is None


def test_staged_ingest_host_half_converts_exactly_or_says_so(built_lib):
    """brie_host_convert_u16 (the host half of the staged count ingest, include/brie_amd.h): integers in [0, 65535]
    become u16 exactly; anything else -- fractional (pseudo-counted input), negative, negative zero, > 65535, NaN, inf --
    is reported, and the slab then travels as the fp32 values themselves."""
    from brie_amd import _capi
    rng = np.random.default_rng(3)
    a = rng.poisson(3.0, (37, 1003)).astype(np.float32)
    a[5, 7], a[36, 1002] = 65535.0, 300.0
    out, bad = _capi.host_convert_u16(a)
    assert not bad and out.dtype == np.uint16
    np.testing.assert_array_equal(out, a.astype(np.uint16))
    for val in (0.5, 0.01, -1.0, -0.0, 65536.0, 1e9, np.nan, np.inf, -np.inf):
        for pos in ((0, 0), (36, 1002), (17, 511)):
            b = a.copy()
            b[pos] = val
            assert _capi.host_convert_u16(b)[1], (val, pos)
    view = a[:, 3:900]                                            # a row pitch that is not the row length
    out, bad = _capi.host_convert_u16(view)
    assert not bad
    np.testing.assert_array_equal(out, view.astype(np.uint16))
    assert _capi.host_convert_u16(np.zeros((0, 5), np.float32))[0].shape == (0, 5)


@pytest.mark.parametrize("dtype", ["float32", "float64", "int32", "int64", "uint8", "uint16", "int16", "uint32"])
def test_typed_ingest_conversion_equals_the_reference_cast(built_lib, dtype):
    """brie_upload_typed's host half (brie_host_convert_slab): a count layer held as integers or float64 is never cast
    by numpy first.  Integers in [0, 65535] travel as exact u16; anything else as the float32 the reference's
    `.astype(np.float32)` (io_utils.py:18, model_wrap.py:111) would give."""
    from brie_amd import _capi
    rng = np.random.default_rng(11)
    hi = {"uint8": 255, "int16": 30000}.get(dtype, 65535)
    a = rng.integers(0, 40, (23, 517)).astype(dtype)
    a[3, 5], a[22, 516] = hi, 1
    out, as_f32 = _capi.host_convert_slab(a)
    assert not as_f32 and out.dtype == np.uint16
    np.testing.assert_array_equal(out, a.astype(np.uint16))
    odd = []                                                    # values that do not fit u16: the float32 cast instead
    if dtype.startswith("float"):
        odd = [0.25, -1.0, 70000.0, np.nan, -0.0, 1e30 if dtype == "float64" else 3e38, 16777217.0]
    elif dtype in ("int32", "int64", "int16"):
        odd = [-1, -30000] + ([70000, 2 ** 31 - 1] if dtype != "int16" else []) + ([2 ** 40 + 1] if dtype == "int64" else [])
    elif dtype == "uint32":
        odd = [65536, 2 ** 32 - 1]
    for val in odd:
        b = a.copy()
        b[11, 200] = val
        out, as_f32 = _capi.host_convert_slab(b)
        assert as_f32 and out.dtype == np.float32, val
        want = b.astype(np.float32)
        assert np.array_equal(out, want, equal_nan=True) and np.array_equal(np.signbit(out), np.signbit(want)), val


def test_non_native_byte_order_layers_take_the_numpy_cast():
    """np.dtype('>i4').name is 'int32' as well: a big-endian layer (HDF5, FITS) handed raw to brie_upload_typed would be
    read as garbage.  Such a layer -- and an unaligned one -- goes through the float32 cast instead."""
    from brie_amd.models import engine
    a = np.arange(24, dtype=">i4").reshape(4, 6)
    out = engine._dense_f32(a)
    assert out.dtype == np.float32 and out.dtype.isnative and np.array_equal(out, np.arange(24).reshape(4, 6))
    native = np.arange(24, dtype=np.int32).reshape(4, 6)
    assert engine._dense_f32(native) is native                  # native integers still go up as they are
    raw = np.zeros(4 * 6 * 4 + 1, np.uint8)[1:].view(np.int32).reshape(4, 6)       # misaligned view
    assert not raw.flags.aligned and engine._dense_f32(raw).dtype == np.float32


def test_unsupported_modes_raise():
    import brie_amd
    with pytest.raises(NotImplementedError):
        brie_amd.BRIE2(10, 10, Kg=1025)
    with pytest.raises(NotImplementedError):
        brie_amd.BRIE2(10, 12, intercept_mode='cell', gene_offset=4)
    m = brie_amd.BRIE2(10, 10)
    with pytest.raises(ValueError):
        m.fit([np.zeros((10, 10))] * 2, target="nonsense")
    with pytest.raises(RuntimeError):
        m.Psi


def test_per_batch_convergence_equals_sequential_reference_batches(patched_wrap):
    """fitBRIE fits all genes concurrently but lets every reference-sized batch stop on its own loss window
    (conv_batch_genes); this must reproduce the sequential per-batch fits of model_wrap.py:241-260."""
    Nc, Ng = 30, 24
    P = make_problem(Nc, Ng, Kc=1, L=2, seed=14, depth=6.0)
    mk = lambda: FakeAnnData({'isoform1': P["counts"][0], 'isoform2': P["counts"][1]})
    kw = dict(Xc=P["Xc"], batch_size=Nc * 8, min_iter=120, max_iter=200, add_iter=10, epsilon_conv=0.1,
              n_loss_gene=3, verbose=False)
    seq = patched_wrap.fitBRIE(mk(), emulate_batches=True, **kw)
    seq_models = list(OracleBackedBRIE2.instances)
    OracleBackedBRIE2.instances = []
    con = patched_wrap.fitBRIE(mk(), **kw)
    (model,) = OracleBackedBRIE2.instances
    n_seq = [m._o.n_iter for m in seq_models]
    assert len(set(n_seq)) > 1, "test data should make the batches stop at different times: %s" % n_seq
    np.testing.assert_array_equal(model._o.n_iter_batch, n_seq)
    np.testing.assert_allclose(con.Psi, seq.Psi, atol=1e-5)
    np.testing.assert_allclose(con.cell_coeff, seq.cell_coeff, atol=1e-5)
    np.testing.assert_allclose(con.sigma, seq.sigma, atol=1e-5)


def test_fitBRIE_missing_unique_layer_raises_keyerror(patched_wrap):
    """model_wrap.py:247,262 index adata.layers[key]: a missing / misspelt unique layer is a KeyError at once; only
    the optional third layer may be absent."""
    P = make_problem(12, 8, Kc=0, L=2, seed=2)
    ad = FakeAnnData({'isoform1': P["counts"][0], 'isoform_2': P["counts"][1]})
    with pytest.raises(KeyError, match="isoform2"):
        patched_wrap.fitBRIE(ad, **FIT)
    ad = FakeAnnData({'isoform1': P["counts"][0], 'isoform2': P["counts"][1]})
    patched_wrap.fitBRIE(ad, **FIT)                                        # no 'ambiguous' layer: fine
    assert ad.uns['brie_param']['layer_keys'] == ['isoform1', 'isoform2']


def test_gene_shard_alignment():
    from brie_amd.sharding import gene_shard
    assert [gene_shard(20000, r, 8, 20) for r in (0, 7)] == [(0, 2500), (17500, 20000)]
    assert gene_shard(26, 1, 2, 8) == (16, 26) and gene_shard(26, 1, 2, 28) == (26, 26)
    with pytest.raises(ValueError):
        gene_shard(26, 0, 2, 6)


def test_every_prototype_of_the_header_is_bound_with_its_arity(built_lib):
    """a signature that changes in include/brie_amd.h without its ctypes binding following links, loads and
    then corrupts the stack.  Every prototype of the header must be exported by the library, be listed in _capi.EXPORTS
    and -- where the binding declares argument types -- take as many arguments as the header says."""
    import re
    from brie_amd import _capi
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "brie_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    protos = re.findall(r"\n(?:int|int64_t|const char \*|void)\s*\*?\s*(brie_\w+)\s*\(([^;{]*?)\)\s*;", hdr)
    assert len(protos) >= 60, len(protos)
    names = [n for n, _ in protos]
    assert sorted(set(names)) == sorted(names)
    missing = sorted(set(names) - set(_capi.EXPORTS))
    extra = sorted(set(_capi.EXPORTS) - set(names))
    assert not missing and not extra, (missing, extra)
    for name, args in protos:
        n = 0 if args.strip() in ("", "void") else len(args.split(","))
        fn = getattr(built_lib, name)
        if fn.argtypes is not None:
            assert len(fn.argtypes) == n, (name, n, len(fn.argtypes))


def test_lrt_models_share_one_device_copy_of_the_counts(monkeypatch):
    """fit_BRIE_matrix hands the count-holding handle from model to model (brie_reconfigure): the layers are uploaded
    once for base + 2 test models, results equal the fresh-model-per-feature run (model_wrap.py:155-187)."""
    import brie_amd.models.wrap as wrap
    from tests.fakes import engine_on_oracle
    Nc, Ng, Kc = 40, 12, 2
    P = make_problem(Nc, Ng, Kc=Kc, L=2, seed=3)
    E = engine_on_oracle()
    monkeypatch.setattr(wrap, "BRIE2", E)
    res = wrap.fit_BRIE_matrix(P["counts"], Xc=P["Xc"], LRT_index=[0, 1], **FIT)
    assert len(E.instances) == 3
    shards = [m._shard for m in E.instances]
    assert shards == [None, None, None]                       # all closed / handed on
    monkeypatch.setattr(wrap, "BRIE2", OracleBackedBRIE2)
    ref = wrap.fit_BRIE_matrix(P["counts"], Xc=P["Xc"], LRT_index=[0, 1], **FIT)
    np.testing.assert_allclose(res.ELBO_gain, ref.ELBO_gain, rtol=1e-5, atol=1e-4)
    np.testing.assert_array_equal(res.Psi, ref.Psi)
    # count the uploads on a second run with an instrumented backend
    E2 = engine_on_oracle()
    made = []
    orig = E2._new_shard

    def counting(self, n_layers):
        sh = orig(self, n_layers)
        made.append(sh)
        return sh
    E2._new_shard = counting
    monkeypatch.setattr(wrap, "BRIE2", E2)
    wrap.fit_BRIE_matrix(P["counts"], Xc=P["Xc"], LRT_index=[0, 1], **FIT)
    assert len(made) == 1 and made[0].uploads == 2            # ONE handle, each layer uploaded once
    # a cell-mode base cannot hand its handle to the gene-mode test models: fresh handles, same answers
    made.clear()
    wrap.fit_BRIE_matrix(P["counts"], Xc=P["Xc"], LRT_index=[0], intercept_mode='cell', **FIT)
    assert len(made) == 2


def test_c_abi_argument_checks_and_loud_failure_without_a_gpu(built_lib):
    """No compute without a GPU: argument validation answers first (BRIE_ERR_INVALID / UNSUPPORTED with a message),
    a well-formed problem fails loudly with BRIE_ERR_HIP on a GPU-less host -- never a CPU fallback."""
    import ctypes
    import torch
    from brie_amd import _capi
    lib = built_lib
    h = ctypes.c_void_p()

    def create(**kw):
        f = dict(abi_version=_capi.ABI_VERSION, device=0, Nc=10, Ng=12, gene_offset=0, Kc=0, Kg=0, n_layers=2, has_efflen=0,
                 intercept_mode=0, train_intercept=1, train_sigma=1, sharded=0, seed=1)
        f.update(kw)
        p = _capi.BrieProblem(*[f[name] for name, _ in _capi.BrieProblem._fields_])
        return lib.brie_create(ctypes.byref(p), ctypes.byref(h)), lib.brie_last_error().decode()
    for kw, code, word in ((dict(Nc=0), -1, "bad shape"), (dict(abi_version=1), -1, "abi_version"), (dict(Kc=1025), -4, "Kc=1025"),
                           (dict(Kg=1025), -4, "Kg=1025"), (dict(n_layers=4), -1, "n_layers"), (dict(n_layers=3), -1, "third count layer"),
                           (dict(gene_offset=6), -1, "multiple of 4"), (dict(intercept_mode=2), -1, "intercept_mode"),
                           (dict(Kg=2, gene_offset=8), -4, "sharded=1")):
        rc, msg = create(**kw)
        assert rc == code and word in msg, (kw, rc, msg)
    assert lib.brie_destroy(None) == 0 and lib.brie_step(None, 1, ctypes.c_float(0.1), 1, None) < 0
    assert lib.brie_comm_init(0, 3, 2, None, ctypes.byref(h)) == -1
    if not torch.cuda.is_available():
        rc, msg = create()
        assert rc == -3 and "hipGetDeviceCount" in msg and not h.value


def test_psi_null_rule_holds_hip_against_a_second_fp32_evaluation():
    """The rule since round 4 (tests/util.py::psi_null_rule) on synthetic runs: `h` = HIP vs the fp32 oracle, `n` = a second
    fp32 CPU evaluation vs the same oracle, both reduced to per-gene summaries first (so that they can be computed where
    the matrices are and judged elsewhere).  What it lets pass and what it refuses; and that slicing the summaries to a
    gene subset equals summarising the sliced matrices."""
    from tests import util
    rng = np.random.default_rng(5)
    Nc, Ng, Kc = 4000, 200, 2
    psi32 = rng.uniform(0.05, 0.95, size=(Nc, Ng)).astype(np.float32)
    base = {"Wc_loc": rng.normal(size=(Kc, Ng)), "intercept": rng.normal(size=Ng), "sigma_log": rng.normal(size=Ng) * 0.1}

    def run(noise=2e-6, displaced=(), clustered=(), scattered=0, shift=2e-3, seed=0):
        r = np.random.default_rng(seed)
        psi = psi32.astype(np.float64) + r.normal(size=psi32.shape) * noise
        par = {k: v.copy() for k, v in base.items()}
        for j in displaced:
            par["intercept"][j] += shift
            psi[:, j] += 0.25 * shift
        for j in clustered:
            psi[r.choice(Nc, Nc // 50, replace=False), j] += 3e-4
        for k in r.choice(Nc * 80, scattered, replace=False):
            psi[k // 80, 120 + k % 80] += 2e-4
        return util.gene_summaries(psi, psi32, par, base)

    rep = util.psi_null_rule(run(seed=1), run(seed=2), "synthetic")
    assert rep["holds"] and rep["displaced_genes"] == {"hip_vs_o32": 0, "o32b_vs_o32": 0} and rep["quiet_genes"]["genes"] == Ng
    assert 5e-6 < rep["quiet_genes"]["p99_upper_bin_edge"]["hip_vs_o32"] < 6e-6        # 2.576 x 2e-6, to one 4.7 % bin
    rep = util.psi_null_rule(run(displaced=(3, 50), clustered=(7,), scattered=20, seed=1),
                             run(displaced=(10,), clustered=(8, 9), scattered=25, seed=2), "synthetic")
    assert rep["displaced_genes"] == {"hip_vs_o32": 2, "o32b_vs_o32": 1}
    assert rep["clustered_genes"]["hip_vs_o32"] == 1 and rep["clustered_genes"]["o32b_vs_o32"] == 2
    assert rep["quiet_genes"]["genes"] == Ng - 6 and rep["quiet_genes"]["gt_1e-4"] == {"hip_vs_o32": 20, "o32b_vs_o32": 25}
    for bad, what in ((dict(displaced=tuple(range(12))), "moved genes"), (dict(clustered=tuple(range(20, 28))), "moved genes"),
                      (dict(scattered=150), "entries beyond 1e-4"), (dict(noise=2.2e-5), "p99"),
                      (dict(displaced=(3,), shift=0.2), "gene shift")):
        with pytest.raises(AssertionError, match=what):
            util.psi_null_rule(run(seed=1, **bad), run(seed=2, scattered=40 if "scattered" in bad else 0), "synthetic")
    rep = util.psi_null_rule(run(seed=1, scattered=150), run(seed=2, scattered=40), "synthetic", check=False)
    assert not rep["holds"] and rep["violated"][0][0].startswith("entries beyond 1e-4")
    # a gene subset of the summaries == the summaries of the gene subset
    full = run(displaced=(3,), clustered=(7,), scattered=30, seed=4)
    cols = np.arange(64)
    sub = util.slice_summaries(full, cols)
    assert sub["hist"].shape == (64, util.NULL_BINS.size) and int(sub["hist"].sum()) == 64 * Nc
    assert np.array_equal(sub["n_gt"], full["n_gt"][:64]) and sub["Nc"] == Nc


def test_psi_ensemble_rule_holds_a_run_against_the_largest_member_and_reports_its_own_false_alarms():
    """The rule since round 5 for the brie-quant default schedule (tests/util.py::psi_ensemble_rule): the statistics of ONE
    comparison are self-contained (comparison_stats), the judged run is held against the LARGEST member value of each with
    the registered constants, and the ensemble's own leave-one-out record is part of the report."""
    import json
    from tests import util
    rng = np.random.default_rng(6)
    Nc, Ng, Kc = 4000, 64, 2
    psi32 = rng.uniform(0.05, 0.95, size=(Nc, Ng)).astype(np.float32)
    base = {"Wc_loc": rng.normal(size=(Kc, Ng)), "intercept": rng.normal(size=Ng), "sigma_log": rng.normal(size=Ng) * 0.1}

    def run(noise=2e-6, displaced=(), clustered=(), scattered=0, shift=2e-3, seed=0):
        r = np.random.default_rng(seed)
        psi = psi32.astype(np.float64) + r.normal(size=psi32.shape) * noise
        par = {k: v.copy() for k, v in base.items()}
        for j in displaced:
            par["intercept"][j] += shift
            psi[:, j] += 0.25 * shift
        for j in clustered:
            psi[r.choice(Nc, Nc // 50, replace=False), j] += 3e-4
        rows = r.choice(Nc, scattered, replace=False)
        for i in range(scattered):                                     # round-robin over genes 24..63: below the cluster size
            psi[rows[i], 24 + i % 40] += 2e-4
        return util.gene_summaries(psi, psi32, par, base)

    members = {"t2": run(displaced=(1, 2), scattered=30, seed=2), "t4": run(displaced=(3,), clustered=(5,), scattered=60, seed=3),
               "t6": run(displaced=(1, 4, 6), scattered=45, seed=4), "x3": run(scattered=20, seed=5)}
    st = util.comparison_stats(members["t4"])
    assert st["displaced_genes"] == 1 and st["clustered_genes"] == 1 and st["moved_genes"] == 2
    assert st["quiet_entries"] == (Ng - 2) * Nc and st["quiet_gt_1e-4"] == 60 and st["shift_max"] == pytest.approx(2e-3)
    rep = util.psi_ensemble_rule(run(displaced=(7, 8, 9), scattered=70, seed=1), members, "synthetic")
    assert rep["holds"] and rep["hip_vs_o32"]["moved_genes"] == 3 and rep["hip_vs_o32"]["quiet_gt_1e-4"] == 70, rep
    assert rep["leave_one_out"]["of"] == 4 and rep["leave_one_out"]["members_failing"] == []
    # largest member: 3 moved genes -> 1.25 x 3 + 2 = 5.75; 60 entries of 62 x 4000 -> 1.25 x 60 + 20 = 95
    for bad, what in ((dict(displaced=tuple(range(10, 16))), "moved_genes"), (dict(scattered=110), "quiet_rate"),
                      (dict(noise=1.2e-5), "quiet_p99"), (dict(displaced=(3,), shift=0.2), "shift_max")):
        r = util.psi_ensemble_rule(run(seed=1, **bad), members, "synthetic", check=False)
        assert not r["holds"] and r["violated"][0][0] == what, (bad, r.get("violated"))
    with pytest.raises(AssertionError):
        util.psi_ensemble_rule(run(seed=1, scattered=110), members, "synthetic")
    # a member that is an outlier of its own ensemble shows up in the leave-one-out record (the rule's false-alarm rate)
    odd = dict(members, t12=run(scattered=160, seed=9))
    r = util.psi_ensemble_rule(run(seed=1), odd, "synthetic")
    assert r["leave_one_out"]["members_failing"] == ["t12"] and r["leave_one_out"]["of"] == 5
    # the registered manifest carries exactly these constants, members and the held-out cases
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    man = json.load(open(os.path.join(root, "tests", "golden", "psi_ensemble_manifest.json")))
    reg = man["registered"]
    assert reg["constants"]["factor"] == util.ENSEMBLE_FACTOR and reg["constants"]["max_factor"] == util.ENSEMBLE_MAX_FACTOR
    assert reg["constants"]["max_floor"] == util.ENSEMBLE_MAX_FLOOR and reg["constants"]["shift_cap"] == util.ENSEMBLE_SHIFT_CAP
    assert sorted(reg["members"]) == ["t12", "t2", "t4", "t6", "t8", "x3"] and reg["genes"] == 64
    assert sorted(k for k, v in reg["cases"].items() if v["held_out"]) == ["c2_cli_64_s5", "c3_cli_64_s5"]
    # entry-level form for the coupled model variants: same constants
    m = {"a": {"n_gt_1e-4": 100, "p99": 2e-5, "max": 1e-3}, "b": {"n_gt_1e-4": 160, "p99": 3e-5, "max": 4e-3}}
    ok = util.entry_ensemble_rule({"n_gt_1e-4": 200, "p99": 4e-5, "max": 5e-3}, m, 104000)
    assert ok["holds"] and ok["leave_one_out"]["members_failing"] == ["b"]
    bad = util.entry_ensemble_rule({"n_gt_1e-4": 230, "p99": 4e-5, "max": 7e-3}, m, 104000, check=False)
    assert [v[0] for v in bad["violated"]] == ["rate", "max"]


def test_psi_parity_rule_counts_moved_genes_as_genes_and_scattered_entries_as_entries():
    """The parity rule itself (tests/util.py::psi_parity_rule, revision 2) on synthetic runs: what it lets pass and what
    it refuses -- so that the rule the GPU tests and bench.py lean on is pinned on the CPU too."""
    from tests import util
    rng = np.random.default_rng(3)
    Nc, Ng, Kc = 4000, 200, 2
    psi64 = rng.uniform(0.05, 0.95, size=(Nc, Ng))
    base = {"Wc_loc": rng.normal(size=(Kc, Ng)), "intercept": rng.normal(size=Ng), "sigma_log": rng.normal(size=Ng) * 0.1}

    def run(noise=2e-6, displaced=(), clustered=(), scattered=0, shift=2e-3, seed=0):
        r = np.random.default_rng(seed)
        psi = psi64 + r.normal(size=psi64.shape) * noise
        par = {k: v.copy() for k, v in base.items()}
        for j in displaced:                                 # an own parameter took another turn: all cells of the gene move
            par["intercept"][j] += shift
            psi[:, j] += 0.25 * shift
        for j in clustered:                                 # parameters agree at the end, 2 % of the cells sit 3e-4 off
            psi[r.choice(Nc, Nc // 50, replace=False), j] += 3e-4
        for k in r.choice(Nc * 80, scattered, replace=False):        # genes 120..199: never moved in these scenarios
            psi[k // 80, 120 + k % 80] += 2e-4
        return psi, par

    def rule(hip, o32):
        return util.psi_parity_rule({"hip": hip[0], "o32": o32[0], "o64": psi64}, {"hip": hip[1], "o32": o32[1], "o64": base}, "synthetic")

    rep = rule(run(seed=1), run(seed=2))
    assert rep["displaced_genes"] == {"hip": 0, "fp32_oracle": 0} and rep["quiet_genes"]["genes"] == Ng
    # gene-level events on either side are exchangeable: other genes, similar numbers
    rep = rule(run(displaced=(3, 50), clustered=(7,), scattered=20, seed=1), run(displaced=(10,), clustered=(8, 9), scattered=25, seed=2))
    assert rep["displaced_genes"] == {"hip": 2, "fp32_oracle": 1} and rep["clustered_genes"]["hip"] == 1 and rep["clustered_genes"]["fp32_oracle"] == 2
    assert rep["quiet_genes"]["genes"] == Ng - 6 and rep["quiet_genes"]["gt_1e-4"] == {"hip": 20, "fp32_oracle": 25}
    # ONE clustered gene carries 80 entries beyond 1e-4: a gene event, not 80 scattered entries (revision 1 refused this)
    rep = rule(run(clustered=(7,), seed=1), run(seed=2))
    assert rep["clustered_genes"]["hip"] == 1 and rep["quiet_genes"]["gt_1e-4"]["hip"] == 0 and rep["undisplaced_genes"]["gt_1e-4"]["hip"] == 80
    # refused: many more moved genes than the reference's own precision produces ...
    with pytest.raises(AssertionError, match="moved genes"):
        rule(run(displaced=tuple(range(0, 12)), seed=1), run(displaced=(100, 101), clustered=(102,), seed=2))
    with pytest.raises(AssertionError, match="moved genes"):
        rule(run(clustered=tuple(range(20, 28)), seed=1), run(seed=2))
    # ... many more scattered entries ...
    with pytest.raises(AssertionError, match="quiet genes"):
        rule(run(scattered=150, seed=1), run(scattered=40, seed=2))
    # ... a worse bulk ...
    with pytest.raises(AssertionError, match="p99"):
        rule(run(noise=2.2e-5, seed=1), run(seed=2))
    # ... one wild entry in an undisplaced gene, or a gene displaced by more than the schedule can explain
    wild = run(seed=1)
    wild[0][5, 5] += 0.01
    with pytest.raises(AssertionError, match="max"):
        rule(wild, run(seed=2))
    with pytest.raises(AssertionError, match="gene shift"):
        rule(run(displaced=(3,), shift=0.2, seed=1), run(displaced=(4,), seed=2))
