"""CPU stand-ins used ONLY by the `-m "not gpu"` host-logic tests.

`OracleBackedBRIE2` has the constructor / fit / attribute surface of
`brie_amd.BRIE2` but runs the CPU oracle, so the host orchestration in
brie_amd/models/wrap.py (LRT bookkeeping, batching, AnnData write-back, gene
sharding) can be exercised without a GPU.  It is monkeypatched into wrap.py by
the tests; the product never imports it.
"""
import numpy as np

from oracle.brie_oracle import OracleBRIE2, add_pseudo_count


class FakeAnnData(object):
    """Duck-typed AnnData: .shape, .layers, .varm, .obsm, .var, .uns."""

    def __init__(self, layers, effLen=None):
        first = next(iter(layers.values()))
        self.shape = first.shape
        self.layers = dict(layers)
        self.varm = {} if effLen is None else {'effLen': effLen}
        self.obsm, self.var, self.uns = {}, {}, {}


class _Arr(np.ndarray):
    def numpy(self):
        return np.asarray(self)


def _w(a):
    return np.asarray(a, np.float32).view(_Arr)


class OracleBackedBRIE2(object):
    instances = []

    def __init__(self, Nc, Ng, Kc=0, Kg=0, effLen=None, intercept=None, intercept_mode='gene',
                 sigma=None, tau_prior=[3, 27], name=None, init_obj=None, seed=0, device=0, gene_offset=0,
                 comm=None):
        self.Nc, self.Ng, self.Kc, self.Kg = Nc, Ng, Kc, Kg
        self.intercept_mode = intercept_mode
        self.Xc = self.Xg = None
        self.seed, self.gene_offset = seed, gene_offset
        self._o = OracleBRIE2(Nc, Ng, Kc, effLen=effLen, intercept=intercept, sigma=sigma, seed=seed,
                              gene_offset=gene_offset, dtype=np.float32, Kg=Kg, intercept_mode=intercept_mode)
        OracleBackedBRIE2.instances.append(self)

    @staticmethod
    def free_device_memory(device=0):
        return 1 << 62

    def fit(self, count_layers, Xc=None, Xg=None, min_iter=1000, max_iter=5000, add_iter=500,
            epsilon_conv=1e-2, verbose=True, n_loss_gene=500, pseudo_count=None, MC_size=1,
            trace_reduce=None, conv_batch_genes=None, **kw):
        self.Xc, self.Xg = Xc, Xg
        data = [np.asarray(c.toarray() if hasattr(c, "toarray") else c, np.float32) for c in count_layers]
        if self._o.effLen is None:
            data = data[:2]
        if pseudo_count:
            data = add_pseudo_count(data, pseudo_count)
        self.fit_args = dict(min_iter=min_iter, max_iter=max_iter, MC_size=MC_size, Kc=self.Kc,
                             Xc=None if Xc is None else np.array(Xc))
        losses = self._o.fit(data, Xc, min_iter, max_iter, add_iter, epsilon_conv, MC_size, n_loss_gene, Xg=Xg,
                             conv_batch_genes=conv_batch_genes)
        if trace_reduce is not None:
            losses = trace_reduce(losses)
        self.losses, self.loss_gene = _w(losses), _w(self._o.loss_gene)
        return self.losses

    def close(self):
        pass

    Z_loc = property(lambda s: _w(s._o.Z_loc))
    Z_std = property(lambda s: _w(s._o.Z_std))
    Psi = property(lambda s: _w(s._o.Psi))
    Psi95CI = property(lambda s: np.asarray(s._o.Psi95CI, np.float32))
    sigma = property(lambda s: _w(s._o.sigma))
    intercept = property(lambda s: _w(s._o.intercept))
    Wc_loc = property(lambda s: _w(s._o.Wc_loc))
    Wg_loc = property(lambda s: _w(s._o.Wg_loc))
