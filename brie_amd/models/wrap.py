"""Host orchestration around `BRIE2`: result object, LRT driver, AnnData front end.

Same names, arguments, defaults and outputs as /root/reference/brie/models/model_wrap.py
(`BRIE_RV` 15-75, `concate` 78-85, `fit_BRIE_matrix` 88-199, `fitBRIE` 202-314).  What is different,
all because the whole gene shard is fitted concurrently on the GPU:
 * genes are not fitted batch after batch; `batch_size` only defines the per-batch convergence groups
   (`conv_batch_genes`), unless `emulate_batches=True` asks for the literal sequential batches;
 * the pseudo-count is applied to the device copy of the counts, the caller's arrays are left untouched
   (the reference mutates them in place, :115-117);
 * with torch.distributed initialised (one process per GPU) genes are sharded over ranks and per-gene
   outputs are all-gathered (brie_amd/sharding.py).
"""
import numpy as np

from .engine import BRIE2
from ..stats import elbo_gain_pval, fdr_bh

verbosity = 3      # brie/settings.py:4
_COMMON_LOSS_DRAW = 0xE0000000      # draw ids of the shared loss_gene evaluation (far above any step counter)

#: result attribute <- model attribute (model_wrap.py:28-38)
_RV_FIELDS = (("sigma", "sigma"), ("intercept", "intercept"), ("cell_coeff", "Wc_loc"), ("gene_coeff", "Wg_loc"),
              ("Psi", "Psi"), ("Psi95CI", "Psi95CI"), ("Z_loc", "Z_loc"), ("Z_std", "Z_std"),
              ("losses", "losses"), ("loss_gene", "loss_gene"))
#: arrays joined along the gene axis by `concate` (model_wrap.py:63-69) and along axis 0 (:71-75)
_GENE_AXIS1 = ("sigma", "intercept", "cell_coeff", "Psi", "Psi95CI", "Z_std", "Z_loc")
_GENE_AXIS0 = ("fdr", "pval", "ELBO_gain")


def _host(x):
    return np.asarray(x.numpy() if hasattr(x, "numpy") else x)


class BRIE_RV(object):
    """Host snapshot of a fitted BRIE2 model (model_wrap.py:15-48)."""

    def __init__(self, model):
        for dim in ("Nc", "Ng", "Kc", "Kg", "Xc", "Xg", "intercept_mode"):
            setattr(self, dim, getattr(model, dim))
        for mine, theirs in _RV_FIELDS:
            setattr(self, mine, _host(getattr(model, theirs)))

    shape = property(lambda self: (self.Nc, self.Ng))
    Wc_loc = property(lambda self: self.cell_coeff)
    Wg_loc = property(lambda self: self.gene_coeff)

    def __str__(self):
        return "BRIE2 results for %d cells and %d genes" % (self.Nc, self.Ng)

    def concate(self, other, axis=1):
        """Append another result along the gene axis (model_wrap.py:53-75); loss traces are laid end to end."""
        if axis != 1:
            print("Warning: only suppoting gene level concate!")
            return None
        self.Ng = self.Ng + other.Ng
        for key in ("losses", "loss_gene"):
            setattr(self, key, np.append(getattr(self, key), getattr(other, key)))
        for key in _GENE_AXIS1:
            setattr(self, key, np.append(getattr(self, key), getattr(other, key), axis=1))
        if hasattr(other, 'ELBO_gain'):
            for key in _GENE_AXIS0:
                setattr(self, key, np.append(getattr(self, key), getattr(other, key), axis=0))


def concate(BRIE_RV_list):
    """Fold a list of per-batch results into the first one (model_wrap.py:78-85)."""
    merged = BRIE_RV_list[0]
    for nxt in BRIE_RV_list[1:]:
        merged.concate(nxt)
    return merged


def _design_for_base(Xc, LRT_index, base_mode, Nc):
    """Features of the base model (model_wrap.py:130-136): all of them ('full'), or all but the tested ones."""
    if base_mode.upper() == 'FULL':
        return Xc.copy()
    if LRT_index is not None and len(LRT_index) < Xc.shape[1]:
        return np.delete(Xc, LRT_index, axis=1)
    return np.ones((Nc, 0), np.float32)


def fit_BRIE_matrix(data, Xc=None, Xg=None, effLen=None, intercept=None, intercept_mode='gene',
                    LRT_index=None, pseudo_count=0.01, sigma=None, base_mode='full',
                    tau_prior=[3, 27], seed=0, device=0, gene_offset=0, comm=None, common_noise=False, **keyargs):
    """Fit a BRIE model with cell / gene features on count matrices (model_wrap.py:88-199).

    common_noise : False = the base and the test models draw independent noise, like the (unseeded) reference.
                   True  = all models of the test share the seed and evaluate their final 500-draw `loss_gene` on the
                   same stretch of the noise stream, so the Monte-Carlo error largely cancels in ELBO_gain
                   (common random numbers; only possible because the noise here is counter-based).

    data : list of 2 or 3 (Nc, Ng) matrices (ndarray, scipy sparse, or torch tensors already in HBM)
    Xc   : (Nc, Kc) float32 cell features;  Xg : (Ng, Kg) float32 gene features
    LRT_index : cell features to test by ELBO gain (None = all, [] = none)
    **keyargs : forwarded to BRIE2.fit (min_iter, max_iter, add_iter, epsilon_conv, MC_size, verbose, ...)
    """
    Nc, Ng = data[0].shape[0], data[0].shape[1]
    print("[BRIE2] adding pseudo_count:", pseudo_count)
    Xc = np.ones((Nc, 0), np.float32) if Xc is None else np.asarray(Xc, np.float32)
    Xg = np.ones((Ng, 0), np.float32) if Xg is None else Xg
    full_base = base_mode.upper() == 'FULL'
    Xc_base = _design_for_base(Xc, LRT_index, base_mode, Nc)

    def run(design, fit_seed, mode=intercept_mode, reuse=None, **extra):
        mdl = BRIE2(Nc=Nc, Ng=Ng, Kc=design.shape[1], Kg=Xg.shape[1], effLen=effLen, intercept=intercept,
                    intercept_mode=mode, sigma=sigma, tau_prior=tau_prior, seed=fit_seed, device=device,
                    gene_offset=gene_offset, comm=comm, reuse=reuse)
        mdl.fit(data, Xc=design, Xg=Xg, pseudo_count=pseudo_count, **dict(fit_args, **extra))
        return mdl

    fit_args = dict(keyargs)
    if common_noise:
        fit_args.setdefault('loss_gene_draw', _COMMON_LOSS_DRAW)

    base = run(Xc_base, seed)
    result = BRIE_RV(base)
    # common noise: the companion fits repeat the base model's stopping times (per batch, or the global one), so that
    # every model has consumed the same stretch of the noise stream when its loss is evaluated
    repeat = {}
    if common_noise:
        sched = getattr(base, 'n_iter_batch', None)
        repeat = dict(n_iter_schedule=np.asarray(sched if sched is not None else [getattr(base, 'n_iter', 0)]))

    tested = np.arange(Xc.shape[1]) if LRT_index is None else LRT_index            # model_wrap.py:149-153
    if len(tested) == 0:
        base.close()
        return result
    # every model of the test is fitted to the same count layers: the device copy (uploaded, pseudo-counted,
    # compacted once) is handed from model to model (brie_reconfigure) instead of being rebuilt per model
    carrier = base

    # ELBO gain per tested feature, in analogy to a likelihood ratio (model_wrap.py:155-187):
    # 'full' base: refit WITHOUT the feature, gain = loss(reduced) - loss(full);
    # 'null' base: refit WITH the feature added, gain = loss(base) - loss(extended), its weight is kept.
    if verbosity == 3 and not common_noise:
        print("[BRIE2] note: ELBO_gain is the difference of two independently sampled loss estimates; "
              "common_noise=True (--commonNoise 1) shares the noise stream and removes most of that error")
    gain = np.zeros((Ng, len(tested)), dtype=np.float32)
    for col, feat in enumerate(tested):
        if verbosity == 3:
            print("[BRIE2] fitting null model without feature %d" % feat if full_base
                  else "[BRIE2] fitting test model by add feature %d" % feat)
        design = np.delete(Xc, feat, 1) if full_base else np.append(Xc_base, Xc[:, feat:(feat + 1)], axis=1)
        # the reference builds these models WITHOUT intercept_mode (model_wrap.py:174-178), i.e. always with
        # the 'gene' default, whatever the base model uses -- mirrored
        # only loss_gene (and the last weight row) of these models is read: no result matrices
        other = run(design, seed if common_noise else seed + 1 + col, mode='gene', reuse=carrier,
                    prefetch_results=False, **repeat)
        carrier.close()
        carrier = other
        other_loss = _host(other.loss_gene)
        if full_base:
            gain[:, col] = other_loss - result.loss_gene
        else:
            gain[:, col] = result.loss_gene - other_loss
            result.cell_coeff = np.append(result.cell_coeff, _host(other.Wc_loc)[-1:, :], axis=0)
    carrier.close()

    result.ELBO_gain = gain
    result.pval = elbo_gain_pval(gain)                                             # chi2.sf(2 gain, 1), :190
    result.fdr = np.stack([fdr_bh(result.pval[:, c]) for c in range(gain.shape[1])], axis=1)   # BH per feature, :193-195
    return result


def _gene_slice(x, g0, g1):
    return x.tocsc()[:, g0:g1] if hasattr(x, "tocsc") else x[:, g0:g1]


def _write_back(adata, res, Xc, Xg, put_layer, LRT_index, params):
    """The AnnData keys of model_wrap.py:272-311."""
    if Xc.shape[0] > 0:
        adata.obsm['Xc'], adata.varm['cell_coeff'] = Xc, res.cell_coeff.T
    if Xg.shape[1] > 0:
        adata.varm['Xg'], adata.obsm['gene_coeff'] = Xg, res.gene_coeff
    per_gene = res.intercept_mode == 'gene'
    per_cell = res.intercept_mode == 'cell'
    if per_gene:
        adata.varm['intercept'] = res.intercept.T
    if per_cell:
        adata.obsm['intercept'], adata.obsm['sigma'] = res.intercept, res.sigma
    else:                                            # 'gene' and every other spelling (e.g. the CLI's "None")
        adata.varm['sigma'] = res.sigma.T
    for key, attr in (('Psi', 'Psi'), ('Z_std', 'Z_std'), ('Psi_95CI', 'Psi95CI')):
        put_layer(key, getattr(res, attr))
    adata.uns['brie_losses'] = res.losses
    adata.var['loss_gene'] = res.loss_gene
    if LRT_index is None or len(LRT_index) >= 1:
        for key in ('fdr', 'pval', 'ELBO_gain'):
            adata.varm[key] = getattr(res, key)
    adata.uns['brie_param'] = params


def _part_bounds(g_lo, g_hi, n_parts, super_batch, n_batch_genes):
    """[g_lo, g_hi) cut into exactly `n_parts` consecutive non-empty pieces, or fewer if that is impossible.
    Pieces of `super_batch` genes when that gives the right count (the rank with the largest shard), otherwise equal
    pieces rounded up to whole gene blocks + convergence batches, whole convergence batches, or gene quads."""
    def cut(piece):
        return [(a, min(a + piece, g_hi)) for a in range(g_lo, g_hi, piece)]
    best = cut(super_batch)
    if len(best) == n_parts:
        return best
    even = -(-(g_hi - g_lo) // n_parts)
    for unit in (int(np.lcm(256, max(1, n_batch_genes))), int(np.lcm(4, max(1, n_batch_genes))), 4):
        parts = cut(-(-even // unit) * unit)
        if len(parts) == n_parts:
            return parts
    return best


def _super_batch_genes(limit, Nc, n_genes, n_layers, Kc, device, n_batch_genes, free=None):
    """Genes per sequential super-batch, or None when the whole range is fitted at once.  'auto': as many genes as
    fit into 90 % of the free HBM (`free`; asked from the device when None); always a multiple of the convergence
    batch and of 256 (one gene block)."""
    from .. import _capi
    unit = int(np.lcm(256, max(1, n_batch_genes)))
    if limit is None:
        return None
    if isinstance(limit, str):
        if limit.lower() != 'auto':
            raise ValueError("max_genes_per_fit=%r" % (limit,))
        if free is None:
            free = BRIE2.free_device_memory(device)
        if _capi.shard_bytes(Nc, n_genes, n_layers, Kc) <= 0.9 * free:
            return None
        per_unit = _capi.shard_bytes(Nc, unit, n_layers, Kc) - (256 << 20)
        limit = int(max(1, (0.9 * free - (256 << 20)) // per_unit)) * unit
    limit = int(limit)
    if limit >= n_genes:
        return None
    return max(unit, limit // unit * unit) if limit >= unit else max(4, limit // 4 * 4)


def fitBRIE(adata, Xc=None, Xg=None, intercept=None, intercept_mode='gene', LRT_index=[],
            layer_keys=['isoform1', 'isoform2', 'ambiguous'], batch_size=500000,
            pseudo_count=0.01, sigma=None, base_mode='full', tau_prior=[3, 27],
            seed=0, device=0, emulate_batches=False, comm=None, gather_layers=True, max_genes_per_fit='auto',
            **keyargs):
    """Fit a BRIE model from an AnnData-like object and write the results back (model_wrap.py:202-314).

    `adata` needs `.shape`, `.layers`, `.varm`, `.obsm`, `.var`, `.uns` (anndata.AnnData,
    brie_amd.io.CountData, or any duck-typed stand-in).  New optional arguments: seed, device,
    emulate_batches, comm (a `brie_amd.sharding.GeneComm`: one process per GPU, genes sharded over
    ranks), gather_layers, max_genes_per_fit ('auto': a gene range that does not fit into the free HBM of the
    device is fitted as sequential super-batches -- what the reference's batch_size does for 500k elements).
    """
    Nc, Ng = adata.shape[0], adata.shape[1]
    Xc = np.ones((Nc, 0), np.float32) if Xc is None else Xc
    Xg = np.ones((Ng, 0), np.float32) if Xg is None else Xg
    LRT_index = np.arange(Xc.shape[1]) if LRT_index is None else LRT_index
    # adata.layers[key] raises KeyError in the reference for a missing unique layer (model_wrap.py:247,262); only the
    # optional third ("ambiguous") layer may be absent
    for key in layer_keys[:2]:
        if key not in adata.layers:
            raise KeyError("count layer %r not in adata.layers (have: %s)" % (key, sorted(adata.layers)))
    layer_keys = list(layer_keys[:2]) + [k for k in layer_keys[2:] if k in adata.layers]
    has_eff = 'effLen' in adata.varm
    # genes are independent unless gene features or per-cell intercepts tie them together (model_wrap.py:241)
    separable = Xg.shape[1] == 0 and intercept_mode.upper() != 'CELL'

    def fit_range(g0, g1, sub_seed):
        layers = [_gene_slice(adata.layers[k], g0, g1) for k in layer_keys]
        return fit_BRIE_matrix(layers, Xc=Xc, Xg=Xg[g0:g1, :], effLen=adata.varm['effLen'][g0:g1, :] if has_eff else None,
                               intercept=intercept, intercept_mode=intercept_mode, LRT_index=LRT_index,
                               pseudo_count=pseudo_count, sigma=sigma, base_mode=base_mode, tau_prior=tau_prior,
                               seed=sub_seed, device=device, gene_offset=g0,
                               comm=None if separable else comm, **keyargs)

    g_lo, g_hi = 0, Ng
    # (a communicator of ONE rank shards nothing; `always_gather` lets a 1-GPU box drive the sharded branch -- trace
    #  all-reduce, end-of-fit gather -- through RCCL anyway: tests/test_gpu_comm.py)
    sharded = comm is not None and (comm.world > 1 or getattr(comm, "always_gather", False))
    n_batch_genes = int(np.ceil(batch_size / Nc))                     # model_wrap.py:242
    ranges = [(0, Ng)]
    if sharded:
        # separable fits shard freely; coupled fits (Kg > 0 / cell mode) shard the genes too and all-reduce the
        # per-cell statistics every step (brie_step_begin/_end).  The loss trace is summed over ranks so that
        # every rank takes the same convergence decisions.
        from ..sharding import gene_shard
        # shard boundaries on whole convergence batches when that leaves no rank empty (else on gene quads: a
        # batch that straddles two ranks is then decided on its windowed loss summed over both, see BRIE2.fit)
        align = int(np.lcm(4, max(1, n_batch_genes))) if separable else 4
        if gene_shard(Ng, comm.world - 1, comm.world, align)[0] >= Ng:
            align = 4
        if gene_shard(Ng, comm.world - 1, comm.world, align)[0] >= Ng:
            raise ValueError("%d genes cannot be sharded over %d ranks (a rank would be empty): use fewer GPUs"
                             % (Ng, comm.world))
        ranges = [gene_shard(Ng, r, comm.world, align) for r in range(comm.world)]
        g_lo, g_hi = ranges[comm.rank]
        # Only the literal independent batch loop (separable AND emulate_batches) exchanges nothing inside the fit.  A
        # coupled model ignores emulate_batches -- it is ONE sharded fit with a per-step all-reduce -- and must take its
        # stopping decisions on the loss summed over ranks like every other sharded fit, or the ranks extend a different
        # number of rounds and their collectives no longer pair up.
        if not (separable and emulate_batches):
            keyargs = dict(keyargs, trace_reduce=comm.allreduce_sum, conv_total_genes=Ng)

    parts = [(g_lo, g_hi)]
    if separable and emulate_batches:                                 # the literal loop of model_wrap.py:244-260
        step = max(4, (n_batch_genes + 3) // 4 * 4)                   # the noise stream is keyed per gene quad
        parts = [(g0, min(g0 + step, g_hi)) for g0 in range(g_lo, g_hi, step)]
    elif separable:
        if 'conv_batch_genes' not in keyargs:
            # all genes at once, but each reference-sized batch still stops on its own loss window
            keyargs = dict(keyargs, conv_batch_genes=n_batch_genes)
        # Larger than the device: sequential super-batches.  With several ranks the split is a COLLECTIVE decision --
        # every fit issues all-reduces, so every rank must run the same number of parts: sized for the largest shard
        # and the smallest free memory, and every rank cuts its own range into that many non-empty pieces.
        n_max = max(b - a for a, b in ranges)
        free = None
        if isinstance(max_genes_per_fit, str) and max_genes_per_fit.lower() == 'auto':
            free = BRIE2.free_device_memory(device)
            if sharded:
                free = int(comm.allreduce_min(np.array([float(free)]))[0])
        super_batch = _super_batch_genes(max_genes_per_fit, Nc, n_max, len(layer_keys), Xc.shape[1], device,
                                         n_batch_genes, free=free)
        if super_batch is not None:
            n_parts = -(-n_max // super_batch)
            cuts = [_part_bounds(a, b, n_parts, super_batch, n_batch_genes) for a, b in ranges]
            short = [r for r, c in enumerate(cuts) if len(c) != n_parts]
            if short:
                raise ValueError("the gene range must be fitted in %d sequential parts, but rank(s) %s hold too few "
                                 "genes to be cut that often: use fewer GPUs or a larger max_genes_per_fit"
                                 % (n_parts, short))
            parts = cuts[comm.rank if sharded else 0]

    if len(parts) == 1:
        ResVal = fit_range(parts[0][0], parts[0][1], seed)
    else:
        done = []
        for g0, g1 in parts:
            done.append(fit_range(g0, g1, seed))
            print("[BRIE2] %d out %d genes done" % (g1, Ng))
        ResVal = concate(done)
        if hasattr(ResVal, 'pval') and not emulate_batches:   # one Benjamini-Hochberg pass over all genes, as unsplit
            ResVal.fdr = np.stack([fdr_bh(ResVal.pval[:, i]) for i in range(ResVal.pval.shape[1])], axis=1)

    ResVal.gene_range = (g_lo, g_hi)
    if sharded:                                                       # RCCL all-gather of the per-gene vectors
        if ResVal.sigma.shape[0] == 1:                                # (Nc,1) cell-mode vectors are replicated
            ResVal.sigma = comm.allgather_genes(ResVal.sigma, Ng, ranges)
            ResVal.intercept = comm.allgather_genes(ResVal.intercept, Ng, ranges)
        ResVal.cell_coeff = comm.allgather_genes(ResVal.cell_coeff, Ng, ranges) if ResVal.cell_coeff.shape[0] \
            else np.zeros((0, Ng), np.float32)
        ResVal.loss_gene = comm.allgather_genes(ResVal.loss_gene, Ng, ranges)[0]
        if hasattr(ResVal, 'ELBO_gain'):
            ResVal.ELBO_gain = comm.allgather_genes(ResVal.ELBO_gain.T, Ng, ranges).T
            ResVal.pval = elbo_gain_pval(ResVal.ELBO_gain)
            ResVal.fdr = np.stack([fdr_bh(ResVal.pval[:, i]) for i in range(ResVal.pval.shape[1])], axis=1)

    whole = (g_lo, g_hi) == (0, Ng)

    def put_layer(key, local):
        """Cell x gene outputs: the full layer when this process holds all genes; otherwise `<key>_shard`
        everywhere and, with gather_layers, the full layer on rank 0 (one gather per layer at the very end)."""
        if whole:
            adata.layers[key] = local
            return
        adata.layers[key + '_shard'] = local
        if gather_layers:
            full = comm.gather_columns(local, Ng, ranges=ranges)
            if full is not None:
                adata.layers[key] = full

    _write_back(adata, ResVal, Xc, Xg, put_layer, LRT_index, {
        'LRT_index': LRT_index, 'base_mode': base_mode, 'intecept': intercept, 'intercept_mode': intercept_mode,
        'sigma': sigma, 'pseudo_count': pseudo_count, 'layer_keys': layer_keys, 'gene_range': (g_lo, g_hi)})
    return ResVal
