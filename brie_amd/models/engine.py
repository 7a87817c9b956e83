"""`BRIE2` -- the reference's model class, backed by HIP kernels on MI355X.

Mirrors the constructor / `fit` / `get_loss` signatures and the attributes of
`brie.models.BRIE2` (/root/reference/brie/models/model_TFProb.py:35-273) so
that `BRIE_RV` and callers written against the reference keep working; the
variational EM loop itself runs in `libbrie_amd.so` through the C ABI of
`include/brie_amd.h`.  There is no CPU path.

Deviations from the reference (documented in DESIGN.md):
 * noise and initial state come from a seeded Philox4x32-10 stream (the
   reference is unseeded); `seed=` is a new optional argument;
 * all genes of a shard are fitted concurrently (the reference loops over
   ~`batch_size/Nc`-gene batches); convergence is decided on the shard's summed
   loss trace, or per reference-sized batch with `conv_batch_genes` (what fitBRIE passes);
 * gene features (`Kg <= 1024`) and `intercept_mode='cell'` couple the genes of a fit: a gene shard then
   needs `comm=` for the per-step all-reduce; `target='marginLik'` works for every model variant.
"""
import time

import numpy as np

from .. import _capi

#: staged learning rates of BRIE2.fit (model_TFProb.py:234)
LEARNING_RATES = (0.001, 0.005, 0.01, 0.02, 0.01, 0.005)


class HostArray(np.ndarray):
    """ndarray that also answers `.numpy()` like the tf tensors BRIE_RV reads
    (model_wrap.py:28-38)."""

    def numpy(self):
        return np.asarray(self)


def _wrap(a):
    return np.asarray(a).view(HostArray)


def _weak(obj):
    import weakref
    try:
        return weakref.ref(obj)
    except TypeError:            # e.g. a list of lists: never considered "the same"
        return None


def _dense_f32(x):
    """logLik_MC densifies sparse layers on the host (model_TFProb.py:135-137); here scipy sparse
    layers are handed to the library as they are and densified on the device (brie_upload_sparse)."""
    if hasattr(x, "tocsc") or hasattr(x, "data_ptr"):
        return x
    if (isinstance(x, np.ndarray) and x.ndim == 2 and x.dtype.name in _capi.TYPED_DTYPES and x.size
            and x.dtype.isnative and x.flags.aligned
            and x.strides[1] == x.itemsize and x.strides[0] % x.itemsize == 0 and x.strides[0] >= x.shape[1] * x.itemsize):
        return x                  # integer / float64 layers (row views included) go up as they are: brie_upload_typed
    return np.ascontiguousarray(x, dtype=np.float32)


class BRIE2(object):
    """
    Ng : number of genes,  Nc : number of cells,
    Kg : number of gene features,  Kc : number of cell features
    (same arguments as model_TFProb.py:42-45, plus seed / device / gene_offset).
    """

    def __init__(self, Nc, Ng, Kc=0, Kg=0, effLen=None, intercept=None, intercept_mode='gene',
                 sigma=None, tau_prior=[3, 27], name=None, init_obj=None,
                 seed=0, device=0, gene_offset=0, comm=None, reuse=None):
        self.Nc, self.Ng, self.Kc, self.Kg = int(Nc), int(Ng), int(Kc), int(Kg)
        self.effLen = effLen                       # (Ng, 3 * 2)
        self.intercept_mode = intercept_mode
        self._intercept_value, self._sigma_value = intercept, sigma
        self._init_obj = init_obj
        self.seed, self.device, self.gene_offset = int(seed), int(device), int(gene_offset)
        self.Xc = self.Xg = None
        self._shard = None
        self._n_layers = None
        self._pseudo_count = None
        self._cell_mode = str(intercept_mode).upper() == 'CELL'          # model_TFProb.py:53-60
        if self.Kg > _capi.MAX_KG:
            raise NotImplementedError("Kg=%d > %d" % (self.Kg, _capi.MAX_KG))
        if self.Kc > _capi.MAX_KC:
            raise NotImplementedError("Kc=%d > %d" % (self.Kc, _capi.MAX_KC))
        # reuse: a fitted model whose device copy of the count layers this model takes over (the models of one
        # likelihood-ratio test are fitted to the same counts, model_wrap.py:155-187) -- see _adopt_shard
        self._reuse = reuse
        self._coupled = self.Kg > 0 or self._cell_mode
        # a gene shard of a coupled fit exchanges per-cell statistics every step through `comm`
        self._comm = comm if (comm is not None and comm.world > 1 and self._coupled) else None
        self._stat = None
        if self._coupled and self.gene_offset != 0 and self._comm is None:
            raise NotImplementedError("gene features / cell intercepts couple all genes: a gene shard needs "
                                      "comm= (brie_amd.sharding.GeneComm) for the per-step all-reduce")

    @staticmethod
    def free_device_memory(device=0):
        """Free HBM of `device` in bytes (fitBRIE sizes its sequential super-batches with it)."""
        return _capi.device_memory(device)[0]

    # ------------------------------------------------------------------ device state
    def _ensure_shard(self, count_layers, Xc, Xg=None):
        n_layers = len(count_layers)
        if self.effLen is None:
            n_layers = 2                          # third layer unused without effLen (model_TFProb.py:162-167)
        if self._shard is not None and self._n_layers == n_layers:
            if not self._same_layers(count_layers[:n_layers]):      # other data on the fitted model (e.g. get_loss)
                self._upload_layers(self._shard, count_layers, n_layers)
            if self.Kc > 0 and Xc is not None:                       # design matrices are taken per call as well
                self._shard.upload(_capi.XC, np.ascontiguousarray(Xc, dtype=np.float32))
            if self.Kg > 0 and Xg is not None:
                self._shard.upload(_capi.XG, np.ascontiguousarray(Xg, dtype=np.float32))
            return self._shard
        sh = self._adopt_shard(count_layers, n_layers)
        if sh is None:
            t0 = time.time()
            sh = self._new_shard(n_layers)
            sh.owner = self
            t1 = time.time()
            self._upload_layers(sh, count_layers, n_layers)
            if hasattr(sh, "synchronize"):
                sh.synchronize()
            # where the ingest goes: allocation of the shard's HBM arrays; host -> device copy of the count layers
            # (staged pipeline for pageable layers, brie_upload) + range check + storage tiers
            self._ingest_timing = {"create_shard_s": t1 - t0, "upload_count_layers_s": time.time() - t1}
            if self.effLen is not None:
                sh.upload(_capi.EFFLEN, np.ascontiguousarray(self.effLen, dtype=np.float32))
        if self.Kc > 0:
            if Xc is None:
                raise ValueError("Kc=%d but Xc is None" % self.Kc)
            sh.upload(_capi.XC, np.ascontiguousarray(Xc, dtype=np.float32))
        if self.Kg > 0:
            if Xg is None:
                raise ValueError("Kg=%d but Xg is None" % self.Kg)
            sh.upload(_capi.XG, np.ascontiguousarray(Xg, dtype=np.float32))
        par = (self.Nc, 1) if self._cell_mode else (1, self.Ng)
        io = self._init_obj
        if io is None:
            sh.init_state(self._intercept_value, self._sigma_value)       # Model_init, model_TFProb.py:12-31
        else:                                                             # init_obj hook, model_TFProb.py:62-84
            sh.init_state(self._intercept_value, self._sigma_value)
            get = (lambda k: io[k]) if isinstance(io, dict) else (lambda k: getattr(io, k))
            sh.upload(_capi.Z_LOC, np.asarray(get('Z_loc'), np.float32))
            if isinstance(io, dict) and 'Z_std_log' in io:
                sh.upload(_capi.Z_STD_LOG, np.asarray(io['Z_std_log'], np.float32))
            else:
                sh.upload(_capi.Z_STD_LOG, np.log(np.asarray(get('Z_std'), np.float32)))
            if self.Kc > 0:
                sh.upload(_capi.WC_LOC, np.asarray(get('Wc_loc'), np.float32).reshape(self.Kc, self.Ng))
            if self.Kg > 0:
                sh.upload(_capi.WG_LOC, np.asarray(get('Wg_loc'), np.float32).reshape(self.Nc, self.Kg))
            sh.upload(_capi.INTERCEPT, np.asarray(get('intercept'), np.float32).reshape(par))
            if isinstance(io, dict) and 'sigma_log' in io:
                sh.upload(_capi.SIGMA_LOG, np.asarray(io['sigma_log'], np.float32).reshape(par))
            else:
                sh.upload(_capi.SIGMA_LOG, np.log(np.asarray(get('sigma'), np.float32)).reshape(par))
        self._shard, self._n_layers = sh, n_layers
        return sh

    def _adopt_shard(self, count_layers, n_layers):
        """Take over the handle of `reuse` when it holds exactly this problem's count layers: the counts stay in HBM
        as uploaded / pseudo-counted / compacted, brie_reconfigure replaces what depends on Kc and the seed."""
        other, self._reuse = self._reuse, None
        sh = getattr(other, "_shard", None)
        if sh is None:
            return None
        if not hasattr(sh, "reconfigure"):
            other.close()
            return None
        same = (other.Nc == self.Nc and other.Ng == self.Ng and other.Kg == self.Kg and other._n_layers == n_layers
                and other._cell_mode == self._cell_mode and other.gene_offset == self.gene_offset
                and other.device == self.device and (other._comm is None) == (self._comm is None)
                and other.effLen is self.effLen and other._pseudo_count == self._pseudo_count
                and other._same_layers(count_layers[:n_layers]))
        if not same:
            # `reuse` hands the earlier model's device memory over either way: when its handle cannot be adopted
            # (other intercept mode, other layers, ...) it is released BEFORE this model allocates its own shard, so
            # two full shards are never resident at once (its results were read by the caller already)
            other.close()
            return None
        other._shard = None                                   # the handle changes owner
        sh.reconfigure(self.Kc, self.seed, self._intercept_value is None, self._sigma_value is None)
        sh.owner = self
        self._layer_refs = other._layer_refs
        return sh

    def _new_shard(self, n_layers):
        """One `brie_handle` (gene shard on one GPU) for this model; the only place a backend is chosen."""
        sh = _capi.Shard(self.Nc, self.Ng, self.Kc, n_layers=n_layers, has_efflen=self.effLen is not None,
                         train_intercept=self._intercept_value is None, train_sigma=self._sigma_value is None,
                         seed=self.seed, device=self.device, gene_offset=self.gene_offset, Kg=self.Kg,
                         intercept_mode=1 if self._cell_mode else 0, sharded=self._comm is not None)
        n_here = getattr(self._comm, "ranks_on_device", 1) if self._comm is not None else 1
        if n_here > 1 and hasattr(sh, "placement_configure"):      # ranks sharing one GPU share its free HBM
            sh.placement_configure(hbm_fraction=0.8 / n_here)
        return sh

    def _upload_layers(self, sh, count_layers, n_layers):
        for l in range(n_layers):
            sh.upload(_capi.COUNT1 + l, _dense_f32(count_layers[l]))
        if self._pseudo_count:
            sh.add_pseudo_count(self._pseudo_count)
        # the device copy is reused while the caller keeps passing the SAME array objects (weak references: the
        # model does not keep the host data alive); arrays modified in place are not detected
        self._layer_refs = [_weak(c) for c in count_layers[:n_layers]]

    def _same_layers(self, layers):
        refs = getattr(self, "_layer_refs", None)
        return refs is not None and len(refs) == len(layers) and all(r is not None and r() is c
                                                                     for r, c in zip(refs, layers))

    def _result(self, which):
        """A cell x gene result matrix: the copy fit() streamed out while loss_gene ran, else a fresh read."""
        cached = getattr(self, "_results", None)
        if cached is not None and which in cached:
            return cached[which]
        return self._need().read(which)

    def _start_result_buffers(self, sh):
        """Host destinations of Psi / Z_std / Psi95CI / Z_loc (what BRIE_RV reads, model_wrap.py:28-35), allocated
        when the fit starts and first-touched by a background thread while the GPU optimises: the page faults of
        16 GB of fresh host memory (configs[2]) leave the critical path.  The buffers stay PAGEABLE: page-locking
        them (brie_host_register) would let the copy engine write at PCIe speed, but registering GBs of user memory
        while kernels run stalls the device queues (measured: +1.0 s on the 996 steps of configs[2],
        profiles/history/r02d_e2e_fit_c3_pinned_while_running.json)."""
        import threading
        bufs = {w: np.empty((self.Nc, self.Ng), np.float32) for w in (_capi.PSI, _capi.Z_STD, _capi.PSI95CI, _capi.Z_LOC)}

        def touch():
            for a in bufs.values():
                a.fill(0)
        th = None
        if getattr(sh, "first_touch_results", False):
            th = threading.Thread(target=touch, daemon=True)
            th.start()
        return bufs, th

    def _need(self):
        if self._shard is None:
            raise RuntimeError("BRIE2 state lives on the GPU and is created by fit()/get_loss()")
        return self._shard

    # ------------------------------------------------------------------ properties (model_TFProb.py:87-116)
    @property
    def Z_loc(self):
        return _wrap(self._result(_capi.Z_LOC))

    @property
    def Z_std_log(self):
        return _wrap(self._need().read(_capi.Z_STD_LOG))

    @property
    def Z_std(self):
        return _wrap(self._result(_capi.Z_STD))

    @property
    def Psi(self):
        """sigmoid(Z_loc) (model_TFProb.py:92-95)"""
        return _wrap(self._result(_capi.PSI))

    @property
    def Psi95CI(self):
        """Width of the 95% interval of the logit-normal posterior (model_TFProb.py:102-106); plain ndarray."""
        return np.asarray(self._result(_capi.PSI95CI))

    @property
    def sigma(self):
        return _wrap(self._need().read(_capi.SIGMA))

    @property
    def sigma_log(self):
        return _wrap(self._need().read(_capi.SIGMA_LOG))

    @property
    def intercept(self):
        return _wrap(self._need().read(_capi.INTERCEPT))

    @property
    def Wc_loc(self):
        return _wrap(self._need().read(_capi.WC_LOC))

    @property
    def Wg_loc(self):
        return _wrap(self._need().read(_capi.WG_LOC))

    # ------------------------------------------------------------------ distribution accessors (model_TFProb.py:97-127)
    @property
    def Z(self):
        """Posterior of the logit Psi: Normal(Z_loc, Z_std) (model_TFProb.py:112-116)."""
        from .dist import Normal
        return Normal(self.Z_loc, self.Z_std)

    @property
    def PsiDist(self):
        """Posterior of Psi: LogitNormal(Z_loc, Z_std) (model_TFProb.py:97-100)."""
        from .dist import LogitNormal
        return LogitNormal(self.Z_loc, self.Z_std)

    @property
    def Z_prior(self):
        """Prior of the logit Psi: Normal(Xc.Wc_loc + Wg_loc.Xg^T + intercept, sigma) (model_TFProb.py:118-127)."""
        from .dist import Normal
        loc = np.zeros((self.Nc, self.Ng), np.float32)
        if self.Kc > 0 and self.Xc is not None:
            loc = loc + np.matmul(np.asarray(self.Xc, np.float32), np.asarray(self.Wc_loc))
        if self.Kg > 0 and self.Xg is not None:
            loc = loc + np.matmul(np.asarray(self.Wg_loc), np.asarray(self.Xg, np.float32).T)
        return Normal(loc + np.asarray(self.intercept), np.asarray(self.sigma))

    def logLik_MC(self, count_layers, target="ELBO", size=10):
        """Monte-Carlo log-likelihood of every (cell, gene) entry, (Nc, Ng) (model_TFProb.py:130-191): mean over
        `size` samples from the posterior (ELBO) or log-mean-exp over samples from the prior (marginLik)."""
        if target not in ("ELBO", "marginLik"):
            raise ValueError("target=%r" % (target,))
        sh = self._ensure_shard(count_layers, self.Xc, self.Xg)
        saved = getattr(self, "target", "ELBO")
        sh.set_target(target)
        try:
            return _wrap(sh.loglik_mc(size))
        finally:
            sh.set_target(saved)

    # ------------------------------------------------------------------ loss (model_TFProb.py:194-211)
    def get_loss(self, count_layers, target="ELBO", axis=None, **kwargs):
        """One stochastic evaluation of the loss (no parameter update), model_TFProb.py:194-211.

        axis=None -> scalar, axis=0 -> per gene (Ng,), axis=1 -> per cell (Nc,); `MC_size` samples.
        target="ELBO": sum KL - sum mean_k ll(z_k); linear in the samples, so it is evaluated as the mean of MC_size
        single-sample evaluations at consecutive draw ids (one read of the data; the noise stream advances by MC_size).
        target="marginLik": -sum log-mean-exp_k ll(z_k), z_k ~ prior (:202-205), the MC_size samples of ONE draw id.
        """
        if target not in ("ELBO", "marginLik"):
            raise ValueError("target=%r" % (target,))
        if axis not in (None, 0, 1):
            raise ValueError("axis=%r (None, 0 or 1: the loss terms are (Nc, Ng) matrices)" % (axis,))
        sh = self._ensure_shard(count_layers, self.Xc, self.Xg)
        mc = int(kwargs.get("MC_size", 1))
        if mc < 1:
            raise ValueError("MC_size=%r" % (mc,))
        saved = getattr(self, "target", "ELBO")
        sh.set_target(target)
        try:
            if axis == 1 or (target == "marginLik" and mc != 1):
                if not hasattr(sh, "get_loss"):
                    raise NotImplementedError("this backend has no per-entry loss accessor")
                lg = sh.get_loss(mc, 1 if axis == 1 else 0)          # per-entry terms, reduced along `axis` (brie_get_loss)
            else:
                lg = sh.loss_gene(mc)                                # one pass, MC_size draws in registers
        finally:
            sh.set_target(saved)
        if axis is None:
            return _wrap(np.float32(lg.astype(np.float64).sum()))
        return _wrap(lg)

    # ------------------------------------------------------------------ fit (model_TFProb.py:214-273)
    def fit(self, count_layers, Xc=None, Xg=None, target="ELBO", optimizer=None, learn_rate=0.05,
            min_iter=1000, max_iter=5000, add_iter=500, epsilon_conv=1e-2, verbose=True,
            n_loss_gene=500, pseudo_count=None, trace_reduce=None, conv_batch_genes=None, loss_gene_draw=None,
            n_iter_schedule=None, conv_total_genes=None, prefetch_results=True, **kwargs):
        """Fit the model's parameters; returns the loss trace like the reference.

        `optimizer` / `learn_rate` are accepted and ignored exactly as in the
        reference (overwritten at model_TFProb.py:228-237).

        conv_batch_genes=None: one model, one convergence decision on the summed loss trace
        (model_TFProb.py:247-258).  conv_batch_genes=n (set by fitBRIE to ceil(batch_size/Nc)): every
        batch of n consecutive genes is one of the reference's sequential fits (model_wrap.py:241-260)
        and stops on its own windowed loss; stopped batches are frozen on the device.  Batches are anchored on the
        GLOBAL gene index (gene_offset + j) // n, so the grouping does not depend on how the genes are sharded; with
        `trace_reduce` and `conv_total_genes` (all genes of the job) the windowed sums of every batch are summed over
        the ranks before the decision, so a batch that straddles two gene shards stops as one.
        n_iter_schedule: iteration counts of an earlier fit (its `n_iter_batch`, or `[n_iter]`) to be repeated
        instead of taking new convergence decisions -- the companion fits of a common-noise LRT stop where the base
        model stopped, so both evaluate the same stretch of the noise stream.
        prefetch_results: stream Psi, Z_std, Psi95CI and Z_loc to the host while the final 500-draw loss_gene pass
        runs (one export pass into pageable destinations that were allocated when the fit started and first-touched
        by a background thread during it); the attributes then return those arrays.  Host memory: four (Nc, Ng)
        float32 arrays are allocated by EVERY fit -- 16 GB at configs[2] -- whether or not the caller reads them;
        pass False when only loss_gene / the per-gene vectors are wanted (fit_BRIE_matrix does for its companion
        models): nothing is read until an attribute is asked for.
        """
        start_time = time.time()
        if target not in ("ELBO", "marginLik"):
            raise ValueError("target=%r" % (target,))
        MC_size = int(kwargs.pop("MC_size", 1))
        if kwargs:
            raise TypeError("unexpected keyword arguments %s" % sorted(kwargs))
        self.Xc, self.Xg, self.target = Xc, Xg, target
        self._pseudo_count = pseudo_count
        sh = self._ensure_shard(count_layers, Xc, Xg)
        sh.set_target(target)
        upload_s = time.time() - start_time                         # host -> device, tiling, count tiers (blocking)
        self._results = None
        staging = self._start_result_buffers(sh) if (prefetch_results and hasattr(sh, "read_results_async")) else None

        native = None
        if self._comm is not None:
            # coupled gene shard: the per-cell statistics are all-reduced every step.  On RCCL the exchange runs
            # INSIDE brie_step on the handle's stream (brie_attach_comm); otherwise (gloo: CPU tests, two ranks on
            # one GPU) through brie_step_begin / torch all-reduce / brie_step_end.
            native = self._comm.native_comm(self.device) if hasattr(self._comm, "native_comm") else None
            if native is not None:
                sh.attach_comm(native)
            elif self._stat is None:
                import torch
                self._stat = torch.zeros(sh.rowstat_size(), dtype=torch.float32,
                                         device=torch.device("cuda", self.device))

        def run(n_steps, lr):
            if self._comm is not None and native is None:
                trace = sh.step_sharded(n_steps, lr, MC_size, self._comm.allreduce_inplace, self._stat)
            else:
                trace = sh.step(n_steps, lr, MC_size)
            if trace_reduce is not None:                             # gene-sharded fit: global loss
                trace = np.asarray(trace_reduce(trace), np.float32)
            return trace

        self.n_iter_batch = None
        losses = np.zeros(0, np.float32)
        stage_s = []
        for i in range(6):                                           # model_TFProb.py:235-241
            t_stage = time.time()
            sh.reset_optimizer()                                     # fresh Adam per stage
            losses = run(int(min_iter / 6), LEARNING_RATES[i])
            stage_s.append(time.time() - t_stage)
        n_iter = min_iter + 0                                        # model_TFProb.py:247-258
        d1 = int(min(50, add_iter / 2))
        d2 = d1 * 2
        if conv_batch_genes and not self._coupled and target == "ELBO":
            cbg = int(conv_batch_genes)
            first = (-self.gene_offset) % cbg                        # first global batch boundary inside this shard
            starts = np.unique(np.concatenate([[0], np.arange(first, self.Ng, cbg)])).astype(np.int64)
            sizes = np.diff(np.append(starts, self.Ng))
            gids = (self.gene_offset + starts) // cbg                # global batch id of every local group
            glob = trace_reduce is not None and conv_total_genes is not None
            n_glob = -(-int(conv_total_genes) // cbg) if glob else 0
            batch_on = np.ones(len(starts), bool)
            self.n_iter_batch = np.full(len(starts), n_iter)
            self.round_log = []                                      # per extension round: where the time goes
            while n_iter < max_iter and len(losses) >= d2 and 0 < d2 <= 128:
                t_round = time.time()
                if n_iter_schedule is not None and len(n_iter_schedule) == len(starts):
                    batch_on &= np.asarray(n_iter_schedule) > n_iter                  # repeat the earlier fit's stops
                else:
                    win = np.add.reduceat(sh.read_loss_window(d2).astype(np.float64), starts, axis=1)
                    if glob:                                         # sums over ALL ranks' genes of every batch
                        full = np.zeros((d2, n_glob))
                        full[:, gids] = win
                        win = np.asarray(trace_reduce(full.ravel()), np.float64).reshape(d2, n_glob)[:, gids]
                    win = win.astype(np.float32)
                    batch_on &= (win[:d1].mean(0) - win[d1:].mean(0)) > epsilon_conv  # per batch, model_TFProb.py:250
                n_on = int(batch_on.sum())
                if trace_reduce is not None:                         # gene shards: stop when no rank has work left
                    n_on = int(round(float(np.asarray(trace_reduce(np.array([float(n_on)])))[0])))
                if n_on == 0:
                    break
                t_dec = time.time()
                sh.set_gene_mask(np.repeat(batch_on, sizes))
                t_mask = time.time()
                n_iter += add_iter
                self.n_iter_batch[batch_on] = n_iter
                losses = np.concatenate([losses, run(add_iter, LEARNING_RATES[5])])
                self.round_log.append({"active_batches": int(batch_on.sum()), "of": len(batch_on),
                                       "decide_s": t_dec - t_round, "mask_pack_s": t_mask - t_dec,
                                       "steps_s": time.time() - t_mask})
            sh.set_gene_mask(None)
            conv_batch_genes = True
        else:
            conv_batch_genes = False
        repeat_to = None if (n_iter_schedule is None or conv_batch_genes) else int(np.max(n_iter_schedule))
        while (not conv_batch_genes and n_iter < max_iter and
               (n_iter < repeat_to if repeat_to is not None else
                (len(losses) >= d2 and d1 > 0 and losses[-d2:-d1].mean() - losses[-d1:].mean() > epsilon_conv))):
            n_iter += add_iter
            losses = np.concatenate([losses, run(add_iter, LEARNING_RATES[5])])

        if loss_gene_draw is not None:       # evaluate the final loss on a FIXED stretch of the noise stream (common
            sh.draw = int(loss_gene_draw)    # random numbers across the models of one LRT, see fit_BRIE_matrix)
        tm = self.timing = {"optimise_s": time.time() - start_time, "of_which_upload_s": upload_s, "stage_s": stage_s}
        if hasattr(sh, "placement_info"):    # which of the probed placements of the streamed arrays the steps ran on, and
            tm["placement"] = sh.placement_info()      # what the search cost (inside stage_s[0]; include/brie_amd.h)
        tm.update(getattr(self, "_ingest_timing", None) or {})
        self._ingest_timing = None
        t0 = time.time()
        if staging is not None:              # results stream out on a second stream while loss_gene computes
            bufs, th = staging
            if th is not None:
                th.join()
            tm["wait_for_first_touch_s"] = time.time() - t0
            sh.read_results_async(bufs[_capi.PSI], bufs[_capi.Z_STD], bufs[_capi.PSI95CI], bufs[_capi.Z_LOC])
        t0 = time.time()
        try:
            self.loss_gene = _wrap(sh.loss_gene(n_loss_gene))        # model_TFProb.py:261-264
        finally:
            # the export worker writes into `bufs`: it is waited for on every path out of here (an exception in
            # loss_gene, KeyboardInterrupt) before the buffers or the handle can go away
            if staging is not None:
                t1 = time.time()
                sh.read_wait()
                tm["read_wait_s"] = time.time() - t1
        tm["loss_gene_s"] = time.time() - t0 - tm.get("read_wait_s", 0.0)
        if staging is not None:
            self._results = bufs
        self.losses = _wrap(losses)
        self.n_iter = n_iter
        if verbose:
            print("[BRIE2] model fit with %d steps in %.2f min, loss: %.2f" % (
                n_iter, (time.time() - start_time) / 60, float(np.sum(self.loss_gene))))
        return self.losses

    def close(self):
        if self._shard is not None:
            self._shard.close()
            self._shard = None
