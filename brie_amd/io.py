"""Minimal count-matrix container and file formats for the brie-quant front end.

`CountData` is a duck-typed stand-in for the slice of `anndata.AnnData` that the
brie-quant path touches (`.X .obs .var .layers .varm .obsm .uns .shape`, boolean
sub-setting on either axis, `.copy()`), because anndata / h5py are not installed in
this image.  When anndata IS importable, `read_h5ad` / `write_h5ad` delegate to it
and everything in brie_amd works on real AnnData objects unchanged.

Formats (from /root/reference/brie/utils/io_utils.py):
 * brie npz (`read_npz`, 55-65): keys `Rmat_dict` {'0','1','2','3'} -> (cells x genes)
   matrices, `effLen_tensor` (genes, 2, 3), `cell_note`, `gene_note` (string tables, first
   row = header, first column = id).  Layer names and `varm['effLen']` =
   [iso1(g1,g2,g3), iso2(g1,g2,g3)] follow `convert_to_annData` (12-52).
 * result table (`dump_results`, 163-199) -> `<out>.brie_ident.tsv`.
"""
import numpy as np
import pandas as pd


class CountData(object):
    def __init__(self, X, obs, var, layers=None, varm=None, obsm=None, uns=None):
        self.X = X
        self.obs, self.var = obs, var
        self.layers = dict(layers or {})
        self.varm, self.obsm, self.uns = dict(varm or {}), dict(obsm or {}), dict(uns or {})

    @property
    def shape(self):
        return (len(self.obs), len(self.var))

    @property
    def n_obs(self):
        return len(self.obs)

    @property
    def n_vars(self):
        return len(self.var)

    def copy(self):
        return self[np.ones(self.n_obs, bool), np.ones(self.n_vars, bool)]

    def __getitem__(self, key):
        rows, cols = key if isinstance(key, tuple) else (key, slice(None))
        ridx = np.arange(self.n_obs)[rows]
        cidx = np.arange(self.n_vars)[cols]

        def cut(m):
            if m is None:
                return None
            m = m[ridx, :] if not hasattr(m, "tocsr") else m.tocsr()[ridx, :]
            return m[:, cidx] if not hasattr(m, "tocsc") else m.tocsc()[:, cidx]
        return CountData(cut(self.X), self.obs.iloc[ridx].copy(), self.var.iloc[cidx].copy(),
                         {k: cut(v) for k, v in self.layers.items()},
                         {k: np.asarray(v)[cidx] for k, v in self.varm.items()},
                         {k: np.asarray(v)[ridx] for k, v in self.obsm.items()}, dict(self.uns))

    def _inplace_subset_var(self, mask):
        """Keep the genes selected by `mask` IN PLACE (the hook filter_genes(copy=False) uses, as on AnnData)."""
        sub = self[:, np.asarray(mask)]
        self.X, self.obs, self.var = sub.X, sub.obs, sub.var
        self.layers, self.varm, self.obsm, self.uns = sub.layers, sub.varm, sub.obsm, sub.uns

    def __repr__(self):
        return "CountData n_obs x n_vars = %d x %d; layers: %s; varm: %s" % (
            self.n_obs, self.n_vars, sorted(self.layers), sorted(self.varm))


def _dense32(m):
    return np.asarray(m.toarray() if hasattr(m, "toarray") else m, dtype=np.float32)


def convert_to_count_data(Rmat_dict, effLen_tensor, cell_note, gene_note, fill_missing=True):
    """io_utils.py:12-52.  Sparse matrices stay sparse (CSC, fp32): the layers are densified on the
    device at upload (brie_upload_sparse), not on the host."""
    import scipy.sparse as sp
    Rmat = {k: (sp.csc_matrix(v, dtype=np.float32) if sp.issparse(v) else _dense32(v)) for k, v in Rmat_dict.items()}
    if fill_missing:
        first = next(iter(Rmat.values()))
        for key in ('0', '1', '2', '3'):
            if key not in Rmat:
                print("key %s not exist in .mtx file, fill with zeros." % key)
                Rmat[key] = sp.csc_matrix(first.shape, dtype=np.float32) if sp.issparse(first) \
                    else np.zeros(first.shape, np.float32)
    layers = {'isoform1': Rmat['1'], 'isoform2': Rmat['2'], 'ambiguous': Rmat['3'], 'poorQual': Rmat['0']}
    cell_note, gene_note = np.asarray(cell_note), np.asarray(gene_note)
    obs = pd.DataFrame(cell_note[1:, :], index=cell_note[1:, 0], columns=cell_note[0, :])
    var = pd.DataFrame(gene_note[1:, :], index=gene_note[1:, 0], columns=gene_note[0, :])
    effLen_tensor = np.asarray(effLen_tensor, np.float32)
    prob = effLen_tensor / effLen_tensor.sum(2, keepdims=True)
    varm = {'effLen': np.append(effLen_tensor[:, 0, :], effLen_tensor[:, 1, :], axis=1),
            'p_ambiguous': prob[:, :, 2]}
    return CountData(Rmat['1'] + Rmat['2'] + Rmat['3'], obs, var, layers, varm)


def read_npz(path):
    """io_utils.py:55-65."""
    dat = np.load(path, allow_pickle=True)
    return convert_to_count_data(dat['Rmat_dict'].item(), dat['effLen_tensor'], dat['cell_note'], dat['gene_note'])


def read_h5ad(path):
    try:
        import anndata
    except ImportError:
        raise ImportError("reading .h5ad needs the `anndata` package, which is not installed; "
                          "use the brie npz format (read_npz) instead")
    return anndata.read_h5ad(path)


def write_results(adata, path):
    """Write the fitted object: .h5ad through anndata when available, otherwise an .npz bundle
    with the same keys (layers/<k>, varm/<k>, obsm/<k>, var/<k>, uns/<k>)."""
    if path.endswith(".h5ad"):
        try:
            import anndata  # noqa: F401
            if hasattr(adata, "write_h5ad"):
                adata.write_h5ad(path)
                return path
        except ImportError:
            pass
        path = path[:-len(".h5ad")] + ".npz"
        print("[BRIE2] anndata not installed: writing %s instead of .h5ad" % path)
    out = {"obs_names": np.asarray(adata.obs.index, str), "var_names": np.asarray(adata.var.index, str)}
    for k, v in adata.layers.items():
        out["layers/" + k] = _dense32(v)
    for k, v in adata.varm.items():
        out["varm/" + k] = np.asarray(v)
    for k, v in adata.obsm.items():
        out["obsm/" + k] = np.asarray(v)
    for k in adata.var.columns:
        out["var/" + k] = np.asarray(adata.var[k])
    for k, v in adata.uns.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                out["uns/%s/%s" % (k, kk)] = np.asarray(vv if vv is not None else "None")
        elif v is not None:
            out["uns/" + k] = np.asarray(v)
    np.savez_compressed(path, **out)
    return path


def dump_results(adata):
    """Splicing-phenotype table (io_utils.py:163-199) as a pandas DataFrame."""
    df = adata.var[['n_counts', 'n_counts_uniq']].copy()
    df['n_counts'] = df['n_counts'].astype(int)
    df['n_counts_uniq'] = df['n_counts_uniq'].astype(int)
    X = adata.X
    df['cdr'] = np.asarray((X > 0).mean(0)).reshape(-1)
    df['intercept'] = adata.varm['intercept'][:, 0] if 'intercept' in adata.varm else [None] * adata.shape[1]
    df['sigma'] = adata.varm['sigma'][:, 0] if 'sigma' in adata.varm else [None] * adata.shape[1]
    LRT_index = adata.uns['brie_param']['LRT_index'] if 'brie_param' in adata.uns else []
    LRT_index = [] if LRT_index is None else LRT_index
    for i in range(len(LRT_index)):
        idx = LRT_index[i]
        if adata.uns.get('Xc_ids') is not None:
            name = str(adata.uns['Xc_ids'][idx])
        else:
            name = 'X%d' % i
        df[name + '_ceoff'] = adata.varm['cell_coeff'][:, i]          # column names as in the reference (sic)
        df[name + '_ELBO_gain'] = adata.varm['ELBO_gain'][:, i]
        df[name + '_pval'] = adata.varm['pval'][:, i]
        df[name + '_FDR'] = adata.varm['fdr'][:, i]
    return df
