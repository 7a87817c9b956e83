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
                 comm=None, reuse=None):
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


class OracleShard(object):
    """The `brie_amd._capi.Shard` surface that `BRIE2.fit` drives, answered by the CPU oracle: lets the CPU suite
    run the engine's REAL control flow (stages, per-batch stopping, gene masks, collectives) without a GPU."""

    m = property(lambda self: self.owner)

    def __init__(self, model, n_layers):
        self.owner, self.n_layers = model, n_layers
        self.uploads = 0
        self.layers = [None] * n_layers
        self.Xc = self.Xg = None
        self.pc = None
        self.target = "ELBO"
        self.o = None

    def upload(self, which, x):
        from brie_amd import _capi
        a = np.asarray(x.toarray() if hasattr(x, "toarray") else x, np.float32)
        if which in (_capi.COUNT1, _capi.COUNT2, _capi.COUNT3):
            self.layers[which - _capi.COUNT1] = a.copy()
            self.uploads += 1
        elif which == _capi.XC:
            self.Xc = a
        elif which == _capi.XG:
            self.Xg = a

    def add_pseudo_count(self, pc):
        self.pc = pc

    def reconfigure(self, Kc, seed, train_intercept=True, train_sigma=True):
        self.Xc, self.o, self.target = None, None, "ELBO"      # counts and pseudo-count stay

    def init_state(self, intercept=None, sigma=None):
        m = self.m
        self.o = OracleBRIE2(m.Nc, m.Ng, m.Kc, effLen=m.effLen, intercept=intercept, sigma=sigma, seed=m.seed,
                             gene_offset=m.gene_offset, dtype=np.float32, Kg=m.Kg, intercept_mode=m.intercept_mode)

    def _data(self):
        return add_pseudo_count(self.layers, self.pc) if self.pc else self.layers

    def set_target(self, target):
        self.target = target

    def reset_optimizer(self):
        self.o.reset_optimizer()

    def step(self, n_steps, lr, mc_size=1, trace=True):
        self.o.Xg = self.Xg
        return np.asarray(self.o.minimize(self._data(), self.Xc, int(n_steps), lr, mc_size, self.target), np.float32)

    def set_gene_mask(self, active=None):
        self.o.gene_active = np.ones(self.m.Ng, bool) if active is None else np.asarray(active, bool)

    def read_loss_window(self, n_last):
        return np.asarray(self.o.lg_hist[-int(n_last):], np.float32)

    def loss_gene(self, n_repeats=500):
        return np.asarray(self.o.eval_loss_gene(self._data(), self.Xc, n_repeats, self.target), np.float32)

    def read(self, which):
        from brie_amd import _capi
        o = self.o
        return np.asarray({_capi.Z_LOC: lambda: o.Z_loc, _capi.Z_STD_LOG: lambda: o.Z_std_log, _capi.Z_STD: lambda: o.Z_std,
                           _capi.PSI: lambda: o.Psi, _capi.PSI95CI: lambda: o.Psi95CI, _capi.SIGMA: lambda: o.sigma,
                           _capi.SIGMA_LOG: lambda: o.sigma_log, _capi.INTERCEPT: lambda: o.intercept,
                           _capi.WC_LOC: lambda: o.Wc_loc, _capi.WG_LOC: lambda: o.Wg_loc}[which](), np.float32)

    draw = property(lambda s: s.o.draw, lambda s, v: setattr(s.o, "draw", int(v)))

    def read_results_async(self, psi=None, z_std=None, psi95ci=None, z_loc=None):
        from brie_amd import _capi
        for dst, which in ((psi, _capi.PSI), (z_std, _capi.Z_STD), (psi95ci, _capi.PSI95CI), (z_loc, _capi.Z_LOC)):
            if dst is not None:
                dst[...] = self.read(which)

    def read_wait(self):
        pass

    def close(self):
        pass


def engine_on_oracle():
    """`brie_amd.BRIE2` (the real host engine) with `OracleShard` as its backend."""
    from brie_amd.models.engine import BRIE2

    class EngineOnOracle(BRIE2):
        instances = []

        def __init__(self, *a, **kw):
            BRIE2.__init__(self, *a, **kw)
            EngineOnOracle.instances.append(self)

        def _new_shard(self, n_layers):
            return OracleShard(self, n_layers)

        @staticmethod
        def free_device_memory(device=0):
            return EngineOnOracle.free_bytes

    EngineOnOracle.free_bytes = 1 << 62
    return EngineOnOracle
