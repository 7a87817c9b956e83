"""Host orchestration around `BRIE2`: result object, LRT driver, AnnData front end.

Same names, arguments, defaults and outputs as
/root/reference/brie/models/model_wrap.py (`BRIE_RV` 15-75, `concate` 78-85,
`fit_BRIE_matrix` 88-199, `fitBRIE` 202-314).  Differences, all because the
whole gene shard is fitted concurrently on the GPU:
 * `batch_size` is accepted and ignored unless `emulate_batches=True`
   (then genes are fitted in reference-sized sequential batches);
 * the pseudo-count is applied to the device copy of the counts, the caller's
   arrays are left untouched (the reference mutates them in place, :115-117);
 * with torch.distributed initialised (one process per GPU) genes are sharded
   over ranks and per-gene outputs are all-gathered (brie_amd/sharding.py).
"""
import numpy as np

from .engine import BRIE2
from ..stats import elbo_gain_pval, fdr_bh

verbosity = 3      # brie/settings.py:4


def _np(x):
    return np.asarray(x.numpy() if hasattr(x, "numpy") else x)


class BRIE_RV(object):
    """Return value object for a fitted BRIE2 model (model_wrap.py:15-48)."""

    def __init__(self, model):
        self.Nc, self.Ng, self.Kc, self.Kg = model.Nc, model.Ng, model.Kc, model.Kg
        self.shape = (self.Nc, self.Ng)
        self.Xc, self.Xg = model.Xc, model.Xg
        self.sigma = _np(model.sigma)
        self.intercept = _np(model.intercept)
        self.cell_coeff = _np(model.Wc_loc)
        self.gene_coeff = _np(model.Wg_loc)
        self.Psi = _np(model.Psi)
        self.Psi95CI = _np(model.Psi95CI)
        self.Z_loc = _np(model.Z_loc)
        self.Z_std = _np(model.Z_std)
        self.losses = _np(model.losses)
        self.loss_gene = _np(model.loss_gene)
        self.intercept_mode = model.intercept_mode

    @property
    def Wc_loc(self):
        return self.cell_coeff

    @property
    def Wg_loc(self):
        return self.gene_coeff

    def __str__(self):
        return "BRIE2 results for %d cells and %d genes" % (self.Nc, self.Ng)

    def concate(self, new_RV, axis=1):
        """Gene-axis concatenation (model_wrap.py:53-75)."""
        if axis != 1:
            print("Warning: only suppoting gene level concate!")
            return None
        self.Ng += new_RV.Ng
        self.shape = (self.Nc, self.Ng)
        self.losses = np.append(self.losses, new_RV.losses)
        self.loss_gene = np.append(self.loss_gene, new_RV.loss_gene)
        for key in ("sigma", "intercept", "cell_coeff", "Psi", "Psi95CI", "Z_std", "Z_loc"):
            setattr(self, key, np.append(getattr(self, key), getattr(new_RV, key), axis=1))
        if hasattr(new_RV, 'ELBO_gain'):
            for key in ("fdr", "pval", "ELBO_gain"):
                setattr(self, key, np.append(getattr(self, key), getattr(new_RV, key), axis=0))


def concate(BRIE_RV_list):
    """Concatenate a list of BRIE results (model_wrap.py:78-85)."""
    res_merge = BRIE_RV_list[0]
    for _res in BRIE_RV_list[1:]:
        res_merge.concate(_res)
    return res_merge


def _n_genes(layer):
    return layer.shape[1]


def fit_BRIE_matrix(data, Xc=None, Xg=None, effLen=None, intercept=None, intercept_mode='gene',
                    LRT_index=None, pseudo_count=0.01, sigma=None, base_mode='full',
                    tau_prior=[3, 27], seed=0, device=0, gene_offset=0, comm=None, **keyargs):
    """Fit a BRIE model with cell features on count matrices (model_wrap.py:88-199).

    data : list of 2 or 3 (Nc, Ng) matrices (ndarray, scipy sparse, or torch tensors in HBM)
    Xc   : (Nc, Kc) float32 cell features;  Xg : (Ng, Kg) -- Kg must be 0
    **keyargs : forwarded to BRIE2.fit (min_iter, max_iter, add_iter, epsilon_conv, MC_size, verbose)
    """
    Nc, Ng = data[0].shape[0], _n_genes(data[0])
    print("[BRIE2] adding pseudo_count:", pseudo_count)
    if Xc is None:
        Xc = np.ones((Nc, 0), np.float32)
    if Xg is None:
        Xg = np.ones((Ng, 0), np.float32)
    Xc = np.asarray(Xc, np.float32)

    if base_mode.upper() == 'FULL':                                   # model_wrap.py:130-136
        Xc_base = Xc.copy()
    elif LRT_index is not None and len(LRT_index) < Xc.shape[1]:
        Xc_base = np.delete(Xc, LRT_index, axis=1)
    else:
        Xc_base = np.ones((Nc, 0), np.float32)

    def run(Xc_fit, fit_seed, mode=intercept_mode):
        mdl = BRIE2(Nc=Nc, Ng=Ng, Kc=Xc_fit.shape[1], Kg=Xg.shape[1], effLen=effLen,
                    intercept=intercept, intercept_mode=mode, sigma=sigma,
                    tau_prior=tau_prior, seed=fit_seed, device=device, gene_offset=gene_offset, comm=comm)
        mdl.fit(data, Xc=Xc_fit, Xg=Xg, pseudo_count=pseudo_count, **keyargs)
        return mdl

    model = run(Xc_base, seed)
    brie_results = BRIE_RV(model)
    model.close()

    if LRT_index is None:                                             # model_wrap.py:149-153
        LRT_index = np.arange(Xc.shape[1])
    if len(LRT_index) == 0:
        return brie_results

    # ELBO gain in analogy to a likelihood ratio (model_wrap.py:155-187)
    ELBO_gain = np.zeros((Ng, len(LRT_index)), dtype=np.float32)
    for ii, idx in enumerate(LRT_index):
        if base_mode.upper() == 'FULL':
            if verbosity == 3:
                print("[BRIE2] fitting null model without feature %d" % (idx))
            Xc_test = np.delete(Xc, idx, 1)
        else:
            if verbosity == 3:
                print("[BRIE2] fitting test model by add feature %d" % (idx))
            Xc_test = np.append(Xc_base, Xc[:, idx:(idx + 1)], axis=1)
        # the reference builds the test models WITHOUT intercept_mode (model_wrap.py:174-178), i.e. always
        # with the 'gene' default, whatever the base model uses -- mirrored
        model_test = run(Xc_test, seed + 1 + ii, mode='gene')
        test_loss_gene = _np(model_test.loss_gene)
        if base_mode.upper() == 'FULL':
            ELBO_gain[:, ii] = test_loss_gene - brie_results.loss_gene
        else:
            ELBO_gain[:, ii] = brie_results.loss_gene - test_loss_gene
            brie_results.cell_coeff = np.append(brie_results.cell_coeff,
                                                _np(model_test.Wc_loc)[-1:, :], axis=0)
        model_test.close()

    brie_results.ELBO_gain = ELBO_gain                                # H1 vs null
    brie_results.pval = elbo_gain_pval(ELBO_gain)                     # model_wrap.py:190
    fdr = np.zeros(ELBO_gain.shape)
    for i in range(fdr.shape[1]):
        fdr[:, i] = fdr_bh(brie_results.pval[:, i])                   # model_wrap.py:193-195
    brie_results.fdr = fdr
    return brie_results


def _gene_slice(x, g0, g1):
    if hasattr(x, "tocsc"):
        return x.tocsc()[:, g0:g1]
    return x[:, g0:g1]


def fitBRIE(adata, Xc=None, Xg=None, intercept=None, intercept_mode='gene', LRT_index=[],
            layer_keys=['isoform1', 'isoform2', 'ambiguous'], batch_size=500000,
            pseudo_count=0.01, sigma=None, base_mode='full', tau_prior=[3, 27],
            seed=0, device=0, emulate_batches=False, comm=None, gather_layers=True, **keyargs):
    """Fit a BRIE model from an AnnData-like object (model_wrap.py:202-314).

    `adata` needs `.shape`, `.layers`, `.varm`, `.obsm`, `.var`, `.uns`
    (anndata.AnnData or any duck-typed stand-in).  Writes the same keys back.
    New optional arguments: seed, device, emulate_batches, comm (a
    `brie_amd.sharding.GeneComm`; one process per GPU, genes sharded over ranks).
    """
    Nc, Ng = adata.shape[0], adata.shape[1]
    if Xc is None:
        Xc = np.ones((Nc, 0), np.float32)
    if Xg is None:
        Xg = np.ones((Ng, 0), np.float32)
    if LRT_index is None:
        LRT_index = np.arange(Xc.shape[1])
    layer_keys = [k for k in layer_keys if k in adata.layers]
    has_eff = 'effLen' in adata.varm
    separable = (Xg is None or Xg.shape[1] == 0) and intercept_mode.upper() != 'CELL'   # model_wrap.py:241

    def fit_range(g0, g1, sub_seed):
        layers = [_gene_slice(adata.layers[k], g0, g1) for k in layer_keys]
        eff = adata.varm['effLen'][g0:g1, :] if has_eff else None
        return fit_BRIE_matrix(layers, Xc=Xc, Xg=Xg[g0:g1, :], effLen=eff, intercept=intercept,
                               intercept_mode=intercept_mode, LRT_index=LRT_index,
                               pseudo_count=pseudo_count, sigma=sigma, base_mode=base_mode,
                               tau_prior=tau_prior, seed=sub_seed, device=device, gene_offset=g0,
                               comm=None if separable else comm, **keyargs)

    g_lo, g_hi = 0, Ng
    if comm is not None and comm.world > 1:
        # separable fits shard freely; coupled fits (Kg > 0 / cell mode) shard the genes too and
        # all-reduce the per-cell statistics every step (brie_step_begin/_end)
        from ..sharding import gene_shard
        g_lo, g_hi = gene_shard(Ng, comm.rank, comm.world)
        # the loss trace is summed over ranks so every rank takes the same convergence decision
        keyargs = dict(keyargs, trace_reduce=comm.allreduce_sum)

    if separable and not emulate_batches and 'conv_batch_genes' not in keyargs:
        # all genes are fitted concurrently; each reference-sized batch still stops on its own loss window
        keyargs = dict(keyargs, conv_batch_genes=int(np.ceil(batch_size / Nc)))
    if separable and emulate_batches:                                 # model_wrap.py:242-260
        _n_gene = int(np.ceil(batch_size / Nc))
        _n_gene = max(4, (_n_gene + 3) // 4 * 4)                      # noise stream is keyed per gene quad
        res_list = []
        for g0 in range(g_lo, g_hi, _n_gene):
            res_list.append(fit_range(g0, min(g0 + _n_gene, g_hi), seed))
            print("[BRIE2] %d out %d genes done" % (min(g0 + _n_gene, g_hi), Ng))
        ResVal = concate(res_list)
    else:
        ResVal = fit_range(g_lo, g_hi, seed)

    ResVal.gene_range = (g_lo, g_hi)
    if comm is not None and comm.world > 1:                           # RCCL all-gather of per-gene vectors
        if ResVal.sigma.shape[0] == 1:                                # (Nc,1) cell-mode vectors are replicated
            ResVal.sigma = comm.allgather_genes(ResVal.sigma, Ng)
            ResVal.intercept = comm.allgather_genes(ResVal.intercept, Ng)
        ResVal.cell_coeff = comm.allgather_genes(ResVal.cell_coeff, Ng) if ResVal.cell_coeff.shape[0] \
            else np.zeros((0, Ng), np.float32)
        ResVal.loss_gene = comm.allgather_genes(ResVal.loss_gene, Ng)[0]
        if hasattr(ResVal, 'ELBO_gain'):
            ResVal.ELBO_gain = comm.allgather_genes(ResVal.ELBO_gain.T, Ng).T
            ResVal.pval = elbo_gain_pval(ResVal.ELBO_gain)
            ResVal.fdr = np.stack([fdr_bh(ResVal.pval[:, i]) for i in range(ResVal.pval.shape[1])], axis=1)

    # update adata (model_wrap.py:272-311).  Sharded fits: rank 0 receives the full cell x gene layers
    # (gather_layers=True, one gather per layer at the very end); other ranks keep `<key>_shard`.
    full = (g_lo, g_hi) == (0, Ng)

    def put_layer(key, local):
        if full:
            adata.layers[key] = local
            return
        adata.layers[key + '_shard'] = local
        if gather_layers:
            whole = comm.gather_columns(local, Ng)
            if whole is not None:
                adata.layers[key] = whole
    if Xc.shape[0] > 0:
        adata.obsm['Xc'] = Xc
        adata.varm['cell_coeff'] = ResVal.cell_coeff.T
    if Xg.shape[1] > 0:
        adata.varm['Xg'] = Xg
        adata.obsm['gene_coeff'] = ResVal.gene_coeff
    if ResVal.intercept_mode == 'gene':
        adata.varm['intercept'] = ResVal.intercept.T
        adata.varm['sigma'] = ResVal.sigma.T
    elif ResVal.intercept_mode == 'cell':
        adata.obsm['intercept'] = ResVal.intercept
        adata.obsm['sigma'] = ResVal.sigma
    else:
        adata.varm['sigma'] = ResVal.sigma.T
    put_layer('Psi', ResVal.Psi)
    put_layer('Z_std', ResVal.Z_std)
    put_layer('Psi_95CI', ResVal.Psi95CI)
    adata.uns['brie_losses'] = ResVal.losses
    adata.var['loss_gene'] = ResVal.loss_gene
    if LRT_index is None or len(LRT_index) >= 1:
        adata.varm['fdr'] = ResVal.fdr
        adata.varm['pval'] = ResVal.pval
        adata.varm['ELBO_gain'] = ResVal.ELBO_gain
    adata.uns['brie_param'] = {
        'LRT_index': LRT_index, 'base_mode': base_mode, 'intecept': intercept,
        'intercept_mode': intercept_mode, 'sigma': sigma, 'pseudo_count': pseudo_count,
        'layer_keys': layer_keys, 'gene_range': (g_lo, g_hi),
    }
    return ResVal
