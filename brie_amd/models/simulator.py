"""Read-count simulator for BRIE2 models on the GPU.

`simulator(adata, ...)` has the call signature and the AnnData side effects of the reference's
`brie.models.simulator.simulator` (/root/reference/brie/models/simulator.py:7-75): it takes the Psi of a
fitted object (or rebuilds it from the fitted prior), turns it into read-category probabilities through the
effective lengths and redraws every (cell, gene) read triple with the observed depth.

Differences, all deliberate:
 * every random number comes from the library's Philox stream (`seed`), on the device, addressed by the global
   (cell, gene) index -- the reference uses unseeded NumPy / TensorFlow samplers;
 * missing inputs raise ValueError instead of print + exit();
 * the gene-feature term of the prior mean is `Wg_loc . Xg^T` read from where fitBRIE stores it
   (`obsm['gene_coeff']`); simulator.py:24 has the operands in an order that only works when Kg == Nc;
 * without effective lengths and with two layers (spliced / unspliced data) the draw is a plain binomial.
"""
import numpy as np
from scipy.special import expit

from .. import _capi

_DEFAULT_LAYERS = ('isoform1', 'isoform2', 'ambiguous')


def _dense(m):
    return np.asarray(m.toarray() if hasattr(m, "toarray") else m, dtype=np.float32)


def prior_mean_logit(adata):
    """Xc.Wc + Wg.Xg^T + intercept(s) from the keys fitBRIE writes (simulator.py:21-28)."""
    mean = np.zeros(adata.shape, np.float32)
    if 'Xc' in adata.obsm and np.shape(adata.obsm['Xc'])[1] > 0:
        mean += np.dot(adata.obsm['Xc'], np.asarray(adata.varm['cell_coeff']).T)
    if 'Xg' in adata.varm and np.shape(adata.varm['Xg'])[1] > 0:
        mean += np.dot(adata.obsm['gene_coeff'], np.asarray(adata.varm['Xg']).T)
    if 'intercept' in adata.varm and np.shape(adata.varm['intercept'])[1] > 0:
        mean += np.asarray(adata.varm['intercept']).T
    if 'intercept' in adata.obsm and np.shape(adata.obsm['intercept'])[1] > 0:
        mean += np.asarray(adata.obsm['intercept'])
    return mean.astype(np.float32)


def simulator(adata, Psi=None, effLen=None, mode="posterior", layer_keys=_DEFAULT_LAYERS, prior_sigma=None,
              seed=0, device=0, gene_offset=0):
    """Simulate read counts for a BRIE model; returns a copy of `adata` with the count layers redrawn.

    mode "posterior": Psi = adata.layers['Psi'];  anything else: Psi ~ logit-normal prior of the fitted
    model (mean from the regression, sd = varm['sigma'] or `prior_sigma`), stored with and without noise in
    layers 'Psi_sim' / 'Psi_sim_noNoise' of the INPUT object, as the reference does."""
    layer_keys = list(layer_keys)
    if Psi is None:
        if "Psi" not in adata.layers:
            raise ValueError("no Psi available in adata.layers")
        if mode == "posterior":
            Psi = _dense(adata.layers['Psi']).copy()
        else:
            mean = prior_mean_logit(adata)
            adata.layers['Psi_sim_noNoise'] = expit(mean)
            if prior_sigma is None:
                sigma = np.asarray(adata.varm['sigma'], np.float32).reshape(-1)
            else:
                sigma = np.full(adata.shape[1], prior_sigma, np.float32)
            Psi = _capi.simulate_psi(mean, sigma, seed=seed, gene_offset=gene_offset, device=device)
    Psi = np.ascontiguousarray(Psi, np.float32)
    adata.layers['Psi_sim'] = Psi

    if effLen is None and 'effLen' in adata.varm and len(layer_keys) == 3:
        effLen = adata.varm['effLen']
    if effLen is None and len(layer_keys) != 2:
        raise ValueError("no effLen available in adata.varm")
    if effLen is not None and len(layer_keys) != 3:
        raise ValueError("effLen given: three layer_keys are needed (or effLen=None with two layers)")

    out = adata.copy()
    total = np.zeros(adata.shape, np.float32)
    for key in layer_keys:
        total += _dense(adata.layers[key])
    counts = _capi.simulate_counts(Psi, total, None if effLen is None else np.asarray(effLen, np.float32),
                                   seed=seed, gene_offset=gene_offset, device=device)
    for key, layer in zip(layer_keys, counts):
        if layer is not None:
            out.layers[key] = layer
    return out
