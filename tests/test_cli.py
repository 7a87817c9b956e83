"""CPU: the brie-quant front end (brie_amd/cli/quant.py, io.py, preprocessing.py) -- flag parity with
/root/reference/brie/bin/quant.py, the reference's filter_genes / match fragments (golden), file formats,
and an end-to-end run through the oracle-backed stand-in engine."""
import os

import numpy as np
import pandas as pd
import pytest
import scipy.sparse as sp

from oracle.synth import make_problem
from tests.fakes import OracleBackedBRIE2

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _count_data(layers):
    from brie_amd.io import CountData
    Nc, Ng = layers['isoform1'].shape
    obs = pd.DataFrame(index=["c%d" % i for i in range(Nc)])
    var = pd.DataFrame(index=["g%d" % i for i in range(Ng)])
    return CountData(sum(layers.values()), obs, var, layers)


def test_filter_genes_and_match_against_reference_fragments():
    from brie_amd.preprocessing import filter_genes, match
    g = np.load(os.path.join(GOLD, "ref_filter_match.npz"), allow_pickle=True)
    layers = {k: g[k] for k in ('isoform1', 'isoform2', 'ambiguous')}
    for case in g["cases"]:
        ad = _count_data({k: v.copy() for k, v in layers.items()})
        out = filter_genes(ad, copy=True, **case["kw"])
        kept = np.isin(ad.var.index, out.var.index)
        np.testing.assert_array_equal(kept, case["kept"])
        np.testing.assert_allclose(out.var['n_counts'].values, case["n_counts"])
        np.testing.assert_allclose(out.var['n_counts_uniq'].values, case["n_counts_uniq"])
        assert out.layers['isoform1'].shape == (ad.shape[0], int(case["kept"].sum()))
    m = match(list(g["ref_ids"]), list(g["new_ids"]))
    got = np.array([-1 if x is None else x for x in m], dtype=int)
    np.testing.assert_array_equal(got, g["match_idx"])


def test_filter_genes_in_place_and_match_duplicates_against_reference():
    """filter_genes(copy=False) filters the caller's object in place and returns None
    (brie/utils/preprocessing.py:37,62,83; used that way at brie/bin/quant.py:70); match() with repeated ref ids
    (brie/utils/base_utils.py:41-59, uniq_ref_only)."""
    from brie_amd.preprocessing import filter_genes, match
    g = np.load(os.path.join(GOLD, "ref_filter_match.npz"), allow_pickle=True)
    d = np.load(os.path.join(GOLD, "ref_match_dup_inplace.npz"), allow_pickle=True)
    ad = _count_data({k: g[k].copy() for k in ('isoform1', 'isoform2', 'ambiguous')})
    names = np.array(ad.var.index)
    ret = filter_genes(ad, copy=False, min_counts=50, min_counts_uniq=10, min_cells_uniq=30)
    want = d["inplace"][0]
    assert ret is None and ad.shape[1] == want["n_left"] and ad.layers['isoform2'].shape[1] == want["n_left"]
    np.testing.assert_array_equal(np.isin(names, ad.var.index), want["kept"])
    np.testing.assert_allclose(ad.var['n_counts'].values, want["n_counts"])

    class Bare(object):
        layers = {k: g[k] for k in ('isoform1', 'isoform2', 'ambiguous')}
        shape = g['isoform1'].shape
    with pytest.raises(TypeError):
        filter_genes(Bare())
    idx = lambda m: np.array([-1 if x is None else x for x in m], dtype=int)
    np.testing.assert_array_equal(idx(match(list(d["dup_ref"]), list(d["dup_new"]))), d["dup_idx"])
    np.testing.assert_array_equal(idx(match(list(d["dup_ref"]), list(d["dup_new"]), uniq_ref_only=False)), d["dup_idx_all"])


def test_parser_defaults_match_reference_cli():
    from brie_amd.cli.quant import build_parser, parse_lrt_index
    o = build_parser().parse_args(["-i", "x.npz"])
    # /root/reference/brie/bin/quant.py:138-187
    assert (o.MC_size, o.min_iter, o.max_iter, o.batch_size, o.pseudo_count, o.nproc) == (3, 5000, 20000, 500000, 0.01, 6)
    assert (o.min_count, o.min_uniq_count, o.min_cell, o.min_MIF) == (50, 10, 30, 0.001)
    assert o.LRT_index == "None" and o.test_base == "full" and o.intercept_mode == "None"
    assert o.layers == "isoform1,isoform2,ambiguous" and o.out_file is None
    assert parse_lrt_index("None") == [] and parse_lrt_index("All") is None
    np.testing.assert_array_equal(parse_lrt_index("0,2"), [0, 2])


def test_main_without_arguments_exits_like_reference(capsys):
    from brie_amd.cli.quant import main
    with pytest.raises(SystemExit) as e:
        main([])
    assert e.value.code == 1
    with pytest.raises(SystemExit) as e:
        main(["--minIter", "10"])
    assert e.value.code == 1 and "need --inFile" in capsys.readouterr().out


def _write_brie_npz(path, P, n_extra_cells=3):
    """A brie-count style npz (io_utils.py:55-65): Rmat_dict keyed '1','2','3', effLen_tensor (Ng,2,3)."""
    Nc, Ng = P["counts"][0].shape
    cells = np.array(["cell%03d" % i for i in range(Nc)])
    genes = np.array(["ENSG%05d" % i for i in range(Ng)])
    cell_note = np.vstack([["cellID"], cells[:, None]])
    gene_note = np.vstack([["GeneID", "GeneName"], np.stack([genes, np.char.add("N", genes)], 1)])
    eff = P["effLen"].reshape(Ng, 2, 3)
    np.savez(path, Rmat_dict={'1': sp.csc_matrix(P["counts"][0]), '2': sp.csc_matrix(P["counts"][1]),
                              '3': sp.csc_matrix(P["counts"][2])},
             effLen_tensor=eff, cell_note=cell_note, gene_note=gene_note)
    return cells, genes


def test_quant_end_to_end_with_oracle_backed_engine(tmp_path, monkeypatch):
    import brie_amd.models.wrap as wrap
    from brie_amd.cli.quant import quant
    from brie_amd.io import read_npz
    monkeypatch.setattr(wrap, "BRIE2", OracleBackedBRIE2)
    Nc, Ng = 50, 30
    P = make_problem(Nc, Ng, Kc=2, L=3, seed=41, depth=8.0)
    in_file = str(tmp_path / "counts.npz")
    cells, genes = _write_brie_npz(in_file, P)
    ad0 = read_npz(in_file)
    assert ad0.shape == (Nc, Ng) and set(ad0.layers) == {'isoform1', 'isoform2', 'ambiguous', 'poorQual'}
    np.testing.assert_allclose(ad0.varm['effLen'], P["effLen"])
    assert sp.issparse(ad0.layers['isoform1'])                  # sparse npz layers stay sparse until the upload
    np.testing.assert_allclose(ad0.X.toarray(), sum(P["counts"]))
    # cell table: shuffled, 5 cells missing, one unknown cell
    perm = np.random.default_rng(0).permutation(Nc)[:-5]
    rows = ["cellID\tgroup\tcov"] + ["%s\t%g\t%g" % (cells[i], P["Xc"][i, 0], P["Xc"][i, 1]) for i in perm]
    rows.append("ghost\t1\t0.5")
    cell_file = str(tmp_path / "cells.tsv")
    open(cell_file, "w").write("\n".join(rows) + "\n")
    out_file = str(tmp_path / "res" / "brie_quant.h5ad")
    ad = quant(in_file, cell_file=cell_file, out_file=out_file, LRT_index=[0], intercept=0, intercept_mode="None",
               min_counts=5, min_counts_uniq=2, min_cells_uniq=3, min_iter=60, max_iter=60, MC_size=1,
               n_loss_gene=3, verbose=False)
    assert ad.shape[0] == Nc - 5                                  # only matched cells are kept, in adata order
    kept_cells = [c for c in cells if c in set(cells[perm])]
    assert list(ad.obs.index) == kept_cells
    np.testing.assert_allclose(ad.obsm['Xc'], P["Xc"][[int(c[4:]) for c in kept_cells]], rtol=1e-5)
    assert ad.shape[1] <= Ng and 'n_counts' in ad.var
    assert 'intercept' not in ad.varm and ad.varm['sigma'].shape == (ad.shape[1], 1)   # interceptMode None
    assert list(ad.uns['Xc_ids']) == ['group', 'cov']
    table = str(tmp_path / "res" / "brie_quant.brie_ident.tsv")
    df = pd.read_csv(table, sep="\t", index_col=0)
    assert list(df.columns) == ['n_counts', 'n_counts_uniq', 'cdr', 'intercept', 'sigma', 'group_ceoff',
                                'group_ELBO_gain', 'group_pval', 'group_FDR']
    assert list(df.index) == list(ad.var.index) and df.index.name == "GeneID"
    np.testing.assert_allclose(df['group_ELBO_gain'].values, ad.varm['ELBO_gain'][:, 0], rtol=2e-3, atol=1e-3)
    bundle = np.load(str(tmp_path / "res" / "brie_quant.npz"), allow_pickle=True)   # no anndata here -> npz bundle
    assert bundle["layers/Psi"].shape == ad.shape and "varm/fdr" in bundle.files and "var/loss_gene" in bundle.files
