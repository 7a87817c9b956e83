"""NumPy restatement of the BRIE2 ELBO hot path.  TEST INFRASTRUCTURE ONLY.

Follows /root/reference/brie/models/model_TFProb.py (cited per function as
`ref:LINE`) with the third-party TF/TFP semantics restated as described in
oracle/__init__.py.  Gradients are hand-derived and are
pinned against torch autograd in tests/test_oracle.py (::test_hand_gradients_match_autograd and its coupled / marginLik twins).

Shapes: counts L x (Nc, Ng); Xc (Nc, Kc); Z_loc, Z_std_log (Nc, Ng);
Wc_loc (Kc, Ng); intercept, sigma_log (1, Ng)  [intercept_mode='gene'].
`dtype` = np.float32 reproduces the reference's precision; np.float64 gives
the precision-independent answer used to bound fp32 effects.
"""
import numpy as np

from . import philox

LEARNING_RATES = (0.001, 0.005, 0.01, 0.02, 0.01, 0.005)    # ref:234
Z975 = 1.959963984540054                                     # ndtri(0.975), ref:105-106 via tfd.LogitNormal.quantile
ADAM_B1, ADAM_B2, ADAM_EPS = 0.9, 0.999, 1e-7                # tf.optimizers.Adam defaults (ref:237)


def log_sigmoid(x):
    """tf.math.log_sigmoid (ref:163-164): -softplus(-x), evaluated stably."""
    return np.minimum(x, 0) - np.log1p(np.exp(-np.abs(x)))


def sigmoid(x):
    """tf.sigmoid (ref:95), evaluated stably in x's dtype."""
    e = np.exp(-np.abs(x))
    return np.where(x >= 0, 1 / (1 + e), e / (1 + e)).astype(x.dtype)


def add_pseudo_count(data, pseudo_count=0.01):
    """model_wrap.py:113-117 -- +pc on BOTH unique layers where c1+c2 > 0 (returns copies)."""
    data = [np.array(d, dtype=np.float32, copy=True) for d in data]
    idx = data[0] + data[1] > 0
    for i in range(2):
        data[i][idx] = data[i][idx] + np.float32(pseudo_count)
    return data


class AdamSlot(object):
    """One Keras-style Adam optimiser variable slot."""

    def __init__(self, shape, dtype):
        self.m = np.zeros(shape, dtype)
        self.v = np.zeros(shape, dtype)


class OracleBRIE2(object):
    """Restatement of `BRIE2` (ref:35-273): gene or cell intercept mode, cell and gene features."""

    def __init__(self, Nc, Ng, Kc=0, effLen=None, intercept=None, sigma=None,
                 seed=0, gene_offset=0, dtype=np.float32, init=None, Kg=0, intercept_mode='gene', variant_b=False):
        self.Nc, self.Ng, self.Kc, self.Kg = int(Nc), int(Ng), int(Kc), int(Kg)
        # variant_b: a SECOND evaluation of the same algorithm in the same precision that differs the way another
        # implementation legitimately may -- Box-Muller evaluated in float32 (within 2e-6 of the fp64-rounded stream) and
        # every reduction over cells / genes taken over the REVERSED axis (other association of the fp32 sums).  The
        # null the HIP path's differences to the fp32 oracle are held against for the model variants the C restatement
        # does not cover (tests/test_gpu_parity.py::test_psi_after_full_default_schedule_model_variants).
        # variant_b = True / 1: both differences; 2: the float32 Box-Muller only; 3: the reversed reductions only; 4: float32
        # Box-Muller + every reduction formed as four contiguous partial sums added in order; 5: the exact noise stream +
        # three partial sums of the reversed axis (five members of the family "another fp32 evaluation": the null ENSEMBLE
        # of the model variants, tests/util.py::entry_ensemble_rule)
        self.variant_b = int(variant_b)
        self._b_noise = self.variant_b in (1, 2, 4)
        self._b_sums = {1: "reverse", 3: "reverse", 4: "blocks4", 5: "blocks3_reverse"}.get(self.variant_b)
        self.cell_mode = str(intercept_mode).upper() == 'CELL'          # ref:53-60
        self.par_shape = (self.Nc, 1) if self.cell_mode else (1, self.Ng)
        self.Xg = None
        self.dtype = np.dtype(dtype)
        self.seed, self.gene_offset = int(seed), int(gene_offset)
        self.effLen = None if effLen is None else np.asarray(effLen, np.float64)
        self.train_intercept = intercept is None        # ref:67-71
        self.train_sigma = sigma is None                # ref:73-78
        self.draw = 0                                   # global noise-draw counter
        dt = self.dtype
        if init is None:
            init = self.model_init(intercept, sigma)
        self.Z_loc = np.array(init['Z_loc'], dt)                    # ref:80
        self.Z_std_log = np.array(init['Z_std_log'], dt)            # ref:82 (log of Z_std)
        self.Wc_loc = np.array(init['Wc_loc'], dt).reshape(self.Kc, self.Ng)    # ref:84
        self.Wg_loc = np.array(init.get('Wg_loc', np.zeros((self.Nc, self.Kg))), dt).reshape(self.Nc, self.Kg)   # ref:85
        self.intercept = np.array(init['intercept'], dt).reshape(self.par_shape)    # ref:67-71
        self.sigma_log = np.array(init['sigma_log'], dt).reshape(self.par_shape)    # ref:73-78
        self.gene_active = np.ones(self.Ng, bool)       # per-batch convergence: frozen genes stop updating
        self.lg_hist = []                               # per-gene losses of the last steps
        self.reset_optimizer()

    # ------------------------------------------------------------------ init
    def model_init(self, intercept=None, sigma=None):
        """`Model_init` (ref:12-31) with the unseeded tf.random.normal replaced by
        the shared Philox stream at draw id INIT_DRAW:
        k=0 Z_loc, k=1 log Z_std (Z_std = exp(N(0,1)), ref:28), k=2 Wc_loc rows
        (cell index = feature index), k=3 intercept (gene mode: cell index 0; cell mode: gene index 0),
        k=4 Wg_loc (gene index = feature index, never offset by the shard)."""
        Nc, Ng, Kc, go = self.Nc, self.Ng, self.Kc, self.gene_offset
        D = philox.INIT_DRAW
        shp = self.par_shape
        out = {
            'Z_loc': philox.normal(self.seed, D, 0, Nc, Ng, go),
            'Z_std_log': philox.normal(self.seed, D, 1, Nc, Ng, go),
            'Wc_loc': philox.normal(self.seed, D, 2, Kc, Ng, go) if Kc > 0
            else np.zeros((0, Ng), np.float32),
        }
        out['Wg_loc'] = philox.normal(self.seed, D, 4, Nc, self.Kg, 0) if self.Kg > 0 \
            else np.zeros((Nc, 0), np.float32)                                    # ref:31
        if intercept is None:                                                     # ref:17-18
            out['intercept'] = philox.normal(self.seed, D, 3, Nc, 1, 0) if self.cell_mode \
                else philox.normal(self.seed, D, 3, 1, Ng, go)
        else:
            out['intercept'] = np.ones(shp, np.float32) * np.float32(intercept)   # ref:20
        if sigma is None:
            out['sigma_log'] = np.zeros(shp, np.float32)                          # log(ones), ref:23,74
        else:
            out['sigma_log'] = np.log(np.ones(shp, np.float32) * np.float32(sigma))   # ref:25,77
        return out

    def reset_optimizer(self):
        """A fresh `tf.optimizers.Adam` (ref:237): zero moments, iteration 0."""
        dt = self.dtype
        self.t = 0
        self.slots = {n: AdamSlot(getattr(self, n).shape, dt)
                      for n in ('Z_loc', 'Z_std_log', 'Wc_loc', 'Wg_loc', 'intercept', 'sigma_log')}

    # ------------------------------------------------------------ properties
    @property
    def Z_std(self):            # ref:88-90
        return np.exp(self.Z_std_log)

    @property
    def sigma(self):            # ref:108-111
        return np.exp(self.sigma_log)

    @property
    def Psi(self):              # ref:92-95  sigmoid(Z_loc)
        return sigmoid(self.Z_loc)

    @property
    def Psi95CI(self):          # ref:102-106
        s = self.Z_std
        return sigmoid(self.Z_loc + self.dtype.type(Z975) * s) - \
            sigmoid(self.Z_loc - self.dtype.type(Z975) * s)

    def prior_mean(self, Xc):   # ref:118-127
        m = np.zeros((self.Nc, self.Ng), self.dtype)
        if self.Kc > 0 and Xc is not None:
            m = np.matmul(np.asarray(Xc, self.dtype), self.Wc_loc)
        if self.Kg > 0 and self.Xg is not None:
            m = m + np.matmul(self.Wg_loc, np.asarray(self.Xg, self.dtype).T)
        return m + self.intercept

    # ------------------------------------------------------------- noise
    def noise(self, MC_size):
        """eps (MC, Nc, Ng) for the current draw id; advances the draw counter."""
        e = np.stack([philox.normal(self.seed, self.draw, k, self.Nc, self.Ng, self.gene_offset,
                                    float_box_muller=self._b_noise)
                      for k in range(MC_size)], axis=0)
        self.draw += 1
        return e.astype(self.dtype)

    # ------------------------------------------------------------- reductions
    def _sum(self, x, axis):
        """x.sum(axis, keepdims=True); variant_b sums the reversed axis and / or in contiguous blocks added in order."""
        if self._b_sums in ("reverse", "blocks3_reverse"):
            x = np.ascontiguousarray(np.flip(x, axis=axis))
        if self._b_sums in ("blocks4", "blocks3_reverse"):
            parts = np.array_split(x, 4 if self._b_sums == "blocks4" else 3, axis=axis)
            out = parts[0].sum(axis=axis, keepdims=True)
            for q in parts[1:]:
                out = out + q.sum(axis=axis, keepdims=True)
            return out
        return x.sum(axis=axis, keepdims=True)

    def _mm(self, a, b):
        """a @ b; variant_b contracts in reversed order and / or in blocks of the contracted axis added in order."""
        if self._b_sums in ("reverse", "blocks3_reverse"):
            a, b = np.ascontiguousarray(a[:, ::-1]), np.ascontiguousarray(b[::-1])
        if self._b_sums in ("blocks4", "blocks3_reverse") and a.shape[1] > 1:
            n = min(a.shape[1], 4 if self._b_sums == "blocks4" else 3)
            cuts = np.array_split(np.arange(a.shape[1]), n)
            out = np.matmul(a[:, cuts[0]], b[cuts[0]])
            for c in cuts[1:]:
                out = out + np.matmul(a[:, c], b[c])
            return out
        return np.matmul(a, b)

    # ------------------------------------------------------------- loss
    def loglik_terms(self, counts, z):
        """Per-sample log-likelihood l(z) and dl/dz  (ref:161-185).

        No effLen (ref:162-167): l = c1 logsig(z) + c2 logsig(-z).
        effLen (ref:168-185): log phi = a - logsumexp(a),
            a = [logsig(z)+log L0, logsig(-z)+log L4, log L5], effLen[:, [0,4,5]] (ref:176).
        """
        dt = self.dtype
        c1, c2 = counts[0], counts[1]
        ls1, ls2 = log_sigmoid(z), log_sigmoid(-z)
        if self.effLen is None:
            ll = c1 * ls1 + c2 * ls2
            g = c1 - (c1 + c2) * sigmoid(z)
            return ll, g
        logL = np.log(self.effLen[:, [0, 4, 5]]).astype(dt)          # (Ng, 3)
        a = np.stack([ls1 + logL[:, 0], ls2 + logL[:, 1],
                      np.zeros_like(z) + logL[:, 2]], axis=-1)
        amax = a.max(axis=-1, keepdims=True)
        lse = amax + np.log(np.exp(a - amax).sum(axis=-1, keepdims=True))
        lphi = a - lse
        ll = c1 * lphi[..., 0] + c2 * lphi[..., 1]
        N = c1 + c2
        if len(counts) > 2:                                           # ref:184-185
            ll = ll + counts[2] * lphi[..., 2]
            N = N + counts[2]
        psi = sigmoid(z)
        phi = np.exp(lphi)
        g = c1 * (1 - psi) - c2 * psi - N * (phi[..., 0] * (1 - psi) - phi[..., 1] * psi)
        return ll.astype(dt), g.astype(dt)

    def margin_loss_and_grads(self, counts, Xc, MC_size=1, eps=None, need_grads=True):
        """`get_loss(target="marginLik")` (ref:156-157,188-189,202-205): z is sampled from the PRIOR
        N(m, sigma), the MC samples are combined with log-mean-exp, there is no KL term and only the
        prior parameters (Wc_loc, Wg_loc, intercept, sigma_log) receive gradients."""
        dt = self.dtype
        counts = [np.asarray(c, dt) for c in counts]
        if eps is None:
            eps = self.noise(MC_size)
        m = self.prior_mean(Xc)
        sig = np.exp(self.sigma_log)
        K = eps.shape[0]
        ll_k = np.zeros((K,) + m.shape, dt)
        g_k = np.zeros((K,) + m.shape, dt)
        for k in range(K):
            z = m + sig * eps[k]                                   # Z_prior.sample (ref:157)
            ll_k[k], g_k[k] = self.loglik_terms(counts, z)
        mx = ll_k.max(axis=0)
        w = np.exp(ll_k - mx)
        ssum = w.sum(axis=0)
        lme = mx + np.log(ssum / dt.type(K))                       # reduce_logmeanexp (ref:189)
        w = w / ssum
        out = {'loss': -lme.sum(), 'loss_gene': -lme.sum(axis=0)}
        if not need_grads:
            return out
        q = (w * g_k).sum(axis=0)                                  # dL/dm = -q
        qe = (w * g_k * eps).sum(axis=0) * sig                     # dL/dlog(sigma) = -qe
        if self.Kc > 0:
            out['Wc_loc'] = -self._mm(np.asarray(Xc, dt).T, q)
        if self.Kg > 0:
            out['Wg_loc'] = -self._mm(q, np.asarray(self.Xg, dt))
        ax = 1 if self.cell_mode else 0
        out['intercept'] = -self._sum(q, ax)
        out['sigma_log'] = -self._sum(qe, ax)
        return out

    def loss_and_grads(self, counts, Xc, MC_size=1, eps=None, need_grads=True):
        """`get_loss(target="ELBO")` (ref:194-211) + its gradient.

        Returns dict with scalar `loss` (= sum KL - sum ll, each reduced
        separately, ref:208-211), per-gene `loss_gene` (axis=0) and gradients.
        """
        dt = self.dtype
        counts = [np.asarray(c, dt) for c in counts]
        if eps is None:
            eps = self.noise(MC_size)
        mu, rho, lam = self.Z_loc, self.Z_std_log, self.sigma_log
        s = np.exp(rho)
        m = self.prior_mean(Xc)
        d = mu - m
        inv_sig2 = np.exp(-2 * lam)
        # tfd.kl_divergence(Normal, Normal) (ref:208; TFP _kl_normal_normal)
        kl = 0.5 * (d * d) * inv_sig2 + 0.5 * np.expm1(2 * (rho - lam)) - (rho - lam)
        ll = np.zeros_like(mu)
        gbar = np.zeros_like(mu)
        gse = np.zeros_like(mu)
        for k in range(eps.shape[0]):
            z = mu + s * eps[k]                     # Normal.sample reparameterisation (ref:159)
            ll_k, g_k = self.loglik_terms(counts, z)
            ll += ll_k
            gbar += g_k
            gse += g_k * s * eps[k]
        K = dt.type(eps.shape[0])
        ll /= K                                      # reduce_mean over MC (ref:191)
        gbar /= K
        gse /= K
        kl_gene = kl.sum(axis=0)
        ll_gene = ll.sum(axis=0)
        out = {'loss': kl.sum() - ll.sum(), 'loss_gene': kl_gene - ll_gene,
               'kl_gene': kl_gene, 'll_gene': ll_gene}
        if not need_grads:
            return out
        r = d * inv_sig2
        s2 = s * s * inv_sig2
        out['Z_loc'] = r - gbar
        out['Z_std_log'] = s2 - 1 - gse
        if self.Kc > 0:
            out['Wc_loc'] = -self._mm(np.asarray(Xc, dt).T, r)
        if self.Kg > 0:
            out['Wg_loc'] = -self._mm(r, np.asarray(self.Xg, dt))
        ax = 1 if self.cell_mode else 0                     # (Nc,1) parameters sum over genes, (1,Ng) over cells
        out['intercept'] = -self._sum(r, ax)
        out['sigma_log'] = self._sum(1 - d * d * inv_sig2 - s2, ax)
        return out

    # ------------------------------------------------------------- Adam
    def trainable(self):
        names = ['Z_loc', 'Z_std_log']
        if self.Kc > 0:
            names.append('Wc_loc')
        if self.Kg > 0:
            names.append('Wg_loc')
        if self.train_intercept:
            names.append('intercept')
        if self.train_sigma:
            names.append('sigma_log')
        return names

    def adam_step(self, grads, lr):
        """Keras Adam `update_step` + variable constraints.
        Only variables that received a gradient are updated (tfp.math.minimize watches the variables
        the loss touches: with target="marginLik" Z_loc / Z_std_log are never read)."""
        dt = self.dtype.type
        self.t += 1
        b1p = np.power(dt(ADAM_B1), dt(self.t))
        b2p = np.power(dt(ADAM_B2), dt(self.t))
        alpha = dt(lr) * np.sqrt(dt(1) - b2p) / (dt(1) - b1p)
        for name in self.trainable():
            if name not in grads:
                continue
            g = grads[name].astype(self.dtype)
            slot = self.slots[name]
            var = getattr(self, name)
            old = (slot.m.copy(), slot.v.copy(), var.copy())
            slot.m += (g - slot.m) * (dt(1) - dt(ADAM_B1))
            slot.v += (g * g - slot.v) * (dt(1) - dt(ADAM_B2))
            var -= (slot.m * alpha) / (np.sqrt(slot.v) + dt(ADAM_EPS))
            if name in ('Z_loc', 'intercept'):          # clip_by_value constraint (ref:69,81)
                np.clip(var, dt(-9), dt(9), out=var)
            if var.shape[-1] == self.Ng and not self.gene_active.all():
                # genes of a finished batch keep state and moments (the reference stopped that fit)
                for arr, o in zip((slot.m, slot.v, var), old):
                    arr[..., ~self.gene_active] = o[..., ~self.gene_active]

    def minimize(self, counts, Xc, num_steps, lr, MC_size=1, target="ELBO"):
        """`tfp.math.minimize` (ref:239-241): trace = loss BEFORE each update."""
        trace = np.zeros(num_steps, self.dtype)
        for i in range(num_steps):
            out = self.loss_and_grads(counts, Xc, MC_size) if target == "ELBO" \
                else self.margin_loss_and_grads(counts, Xc, MC_size)
            trace[i] = out['loss']
            lg = np.asarray(out['loss_gene'], np.float64)
            if self.lg_hist and not self.gene_active.all():       # frozen genes carry their last loss forward
                lg = np.where(self.gene_active, lg, self.lg_hist[-1])
                trace[i] = lg.sum()
            self.lg_hist = (self.lg_hist + [lg])[-128:]
            self.adam_step(out, lr)
        return trace

    # ------------------------------------------------------------- fit
    def fit(self, counts, Xc=None, min_iter=1000, max_iter=5000, add_iter=500,
            epsilon_conv=1e-2, MC_size=1, n_loss_gene=500, Xg=None, target="ELBO", conv_batch_genes=None):
        """`BRIE2.fit` (ref:214-273).

        conv_batch_genes=None: one model, one convergence decision on the summed trace (ref:247-258).
        conv_batch_genes=n: the fitBRIE situation (model_wrap.py:241-260) -- every batch of n consecutive
        genes is its own reference fit and stops on its own windowed loss; a stopped batch is frozen."""
        self.Xc = Xc
        if Xg is not None:
            self.Xg = Xg
        for i in range(6):                                   # ref:235-241
            self.reset_optimizer()
            losses = self.minimize(counts, Xc, int(min_iter / 6), LEARNING_RATES[i], MC_size, target)
        n_iter = min_iter + 0                                # ref:247
        d1 = int(min(50, add_iter / 2))                      # ref:248
        d2 = d1 * 2
        if conv_batch_genes:
            starts = np.arange(0, self.Ng, int(conv_batch_genes))
            batch_on = np.ones(len(starts), bool)
            self.n_iter_batch = np.full(len(starts), n_iter)
            while n_iter < max_iter and len(self.lg_hist) >= d2 and d1 > 0:
                win = np.add.reduceat(np.asarray(self.lg_hist[-d2:]), starts, axis=1).astype(np.float32)
                batch_on &= (win[:d1].mean(0) - win[d1:].mean(0)) > epsilon_conv      # ref:250, per batch
                if not batch_on.any():
                    break
                self.gene_active = np.repeat(batch_on, np.diff(np.append(starts, self.Ng)))
                n_iter += add_iter
                self.n_iter_batch[batch_on] = n_iter
                losses = np.concatenate([losses, self.minimize(
                    counts, Xc, add_iter, LEARNING_RATES[5], MC_size, target)])
            self.gene_active = np.ones(self.Ng, bool)
        while (not conv_batch_genes and losses[-d2:-d1].mean() - losses[-d1:].mean() > epsilon_conv
               and n_iter < max_iter):                       # ref:250-251
            n_iter += add_iter
            losses = np.concatenate([losses, self.minimize(
                counts, Xc, add_iter, LEARNING_RATES[5], MC_size, target)])
        self.n_iter = n_iter
        self.loss_gene = self.eval_loss_gene(counts, Xc, n_loss_gene, target)
        self.losses = losses
        return losses

    def eval_loss_gene(self, counts, Xc, n_repeats=500, target="ELBO"):
        """ref:261-264 -- mean of `n_repeats` stochastic get_loss(axis=0), MC_size=1
        (the call at ref:261 does not forward **kwargs)."""
        acc = np.zeros(self.Ng, self.dtype)
        fn = self.loss_and_grads if target == "ELBO" else self.margin_loss_and_grads
        for _ in range(n_repeats):
            acc += fn(counts, Xc, 1, need_grads=False)['loss_gene']
        return acc / self.dtype.type(n_repeats)
