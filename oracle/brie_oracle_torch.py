"""Eager torch-CPU autograd restatement of BRIE2.  TEST INFRASTRUCTURE ONLY.

Purpose 1 -- pin the hand-derived gradients of oracle/brie_oracle.py against
an automatic-differentiation run of a line-by-line restatement of
`BRIE2.get_loss` (/root/reference/brie/models/model_TFProb.py:118-211).

Purpose 2 -- `bench.py`'s `cpu_baseline` (kind "port"): the reference's
execution shape -- an eager per-op tensor library in fp32,
materialised (MC, Nc, Ng) temporaries, tape/autograd backward, one Adam update
per variable, and the reference's gene batching
`n_gene = ceil(batch_size / Nc)` (model_wrap.py:241-243) -- because
TensorFlow itself is not installed and the reference's Python cannot travel to
the GPU box.  Label: "reference-equivalent CPU path (restated; TF absent)".
"""
import math

import numpy as np
import torch

from . import philox
from .brie_oracle import LEARNING_RATES, ADAM_B1, ADAM_B2, ADAM_EPS


class TorchBRIE2(object):
    """`BRIE2` (ref model_TFProb.py:35-273) on torch tensors, gene mode, Kg = 0."""

    def __init__(self, Nc, Ng, Kc=0, effLen=None, intercept=None, sigma=None,
                 init=None, seed=0, gene_offset=0, dtype=torch.float32, noise='philox', Kg=0,
                 intercept_mode='gene'):
        self.Nc, self.Ng, self.Kc, self.Kg = Nc, Ng, Kc, Kg
        par_shape = (Nc, 1) if str(intercept_mode).upper() == 'CELL' else (1, Ng)      # ref:53-60
        self.Xg = None
        self.dtype = dtype
        self.seed, self.gene_offset, self.draw = seed, gene_offset, 0
        self.noise_mode = noise
        self.effLen = None if effLen is None else torch.as_tensor(np.asarray(effLen), dtype=dtype)
        if init is None:                                        # Model_init, ref:12-31
            g = torch.Generator().manual_seed(seed)
            init = {
                'Z_loc': torch.randn(Nc, Ng, generator=g),
                'Z_std_log': torch.randn(Nc, Ng, generator=g),
                'Wc_loc': torch.randn(Kc, Ng, generator=g),
                'Wg_loc': torch.randn(Nc, Kg, generator=g),
                'intercept': torch.randn(*par_shape, generator=g) if intercept is None
                else torch.ones(*par_shape) * intercept,
                'sigma_log': torch.zeros(*par_shape) if sigma is None
                else torch.log(torch.ones(*par_shape) * sigma),
            }

        def var(x, train):
            return torch.tensor(np.asarray(x), dtype=dtype).reshape(np.asarray(x).shape).clone() \
                .requires_grad_(train)
        self.Z_loc = var(init['Z_loc'], True)                   # ref:80
        self.Z_std_log = var(init['Z_std_log'], True)           # ref:82
        self.Wc_loc = var(np.asarray(init['Wc_loc']).reshape(Kc, Ng), Kc > 0)   # ref:84
        self.Wg_loc = var(np.asarray(init.get('Wg_loc', np.zeros((Nc, Kg)))).reshape(Nc, Kg), Kg > 0)   # ref:85
        self.intercept = var(np.asarray(init['intercept']).reshape(par_shape), intercept is None)   # ref:67-71
        self.sigma_log = var(np.asarray(init['sigma_log']).reshape(par_shape), sigma is None)       # ref:73-78
        self.Xc = None

    # -- ref:118-127
    def Z_prior_loc(self):
        zz = torch.zeros((self.Nc, self.Ng), dtype=self.dtype)
        if self.Kc > 0 and self.Xc is not None:
            zz = torch.matmul(self.Xc, self.Wc_loc)
        if self.Kg > 0 and self.Xg is not None:
            zz = zz + torch.matmul(self.Wg_loc, self.Xg.T)                       # ref:124-125
        return zz + self.intercept

    def _eps(self, MC_size):
        if self.noise_mode == 'philox':
            e = np.stack([philox.normal(self.seed, self.draw, k, self.Nc, self.Ng, self.gene_offset)
                          for k in range(MC_size)], 0)
            self.draw += 1
            return torch.as_tensor(e).to(self.dtype)
        return torch.randn(MC_size, self.Nc, self.Ng, dtype=self.dtype)

    # -- ref:130-191 (target="ELBO")
    def logLik_MC(self, count_layers, MC_size=1, eps=None):
        if eps is None:
            eps = self._eps(MC_size)
        _Z = self.Z_loc.unsqueeze(0) + torch.exp(self.Z_std_log).unsqueeze(0) * eps   # Normal.sample, ref:159
        return torch.mean(self._loglik(count_layers, _Z), dim=0)                      # ref:191

    def _loglik(self, count_layers, _Z):
        """Element-wise log-likelihood of the sampled logits (ref:161-185) -> (MC, Nc, Ng)."""
        if self.effLen is None:                                  # ref:162-167
            Psi1_log = torch.nn.functional.logsigmoid(_Z)
            Psi2_log = torch.nn.functional.logsigmoid(0 - _Z)
            _logLik_S = (count_layers[0].unsqueeze(0) * Psi1_log +
                         count_layers[1].unsqueeze(0) * Psi2_log)
        else:                                                    # ref:168-185
            _Z = _Z.unsqueeze(3)
            Psi_logs = torch.cat((torch.nn.functional.logsigmoid(_Z),
                                  torch.nn.functional.logsigmoid(0 - _Z),
                                  torch.zeros_like(_Z)), dim=3)
            effLen = self.effLen[:, [0, 4, 5]].unsqueeze(0).unsqueeze(0)
            phi_log = Psi_logs + torch.log(effLen)
            phi_log = phi_log - torch.logsumexp(phi_log, dim=3, keepdim=True)
            _logLik_S = (count_layers[0].unsqueeze(0) * phi_log[:, :, :, 0] +
                         count_layers[1].unsqueeze(0) * phi_log[:, :, :, 1])
            if len(count_layers) > 2:
                _logLik_S = _logLik_S + count_layers[2].unsqueeze(0) * phi_log[:, :, :, 2]
        return _logLik_S

    # -- ref:202-205 (target="marginLik"): z ~ prior, log-mean-exp over the MC axis, no KL
    def get_margin_loss(self, count_layers, axis=None, MC_size=1, eps=None):
        if eps is None:
            eps = self._eps(MC_size)
        _Z = self.Z_prior_loc().unsqueeze(0) + torch.exp(self.sigma_log).unsqueeze(0) * eps      # ref:157
        ll = self._loglik(count_layers, _Z)
        lme = torch.logsumexp(ll, dim=0) - math.log(ll.shape[0])                                 # ref:189
        return -torch.sum(lme) if axis is None else -torch.sum(lme, dim=axis)

    # -- ref:194-211 (target="ELBO")
    def get_loss(self, count_layers, axis=None, MC_size=1, eps=None):
        loc_q, loc_p = self.Z_loc, self.Z_prior_loc()
        # TFP _kl_normal_normal: 0.5*((a.loc-b.loc)/b.scale)^2 + 0.5*expm1(2*dlog) - dlog
        diff_log_scale = self.Z_std_log - self.sigma_log
        kl = (0.5 * ((loc_q - loc_p) / torch.exp(self.sigma_log)) ** 2 +
              0.5 * torch.expm1(2.0 * diff_log_scale) - diff_log_scale)
        ll = self.logLik_MC(count_layers, MC_size, eps)
        if axis is None:
            return torch.sum(kl) - torch.sum(ll)
        return torch.sum(kl, dim=axis) - torch.sum(ll, dim=axis)

    def variables(self):
        return [v for v in (self.Z_loc, self.Z_std_log, self.Wc_loc, self.Wg_loc, self.intercept, self.sigma_log)
                if v.requires_grad]

    def new_adam(self, lr):
        """tf.optimizers.Adam state (ref:237)."""
        return {'lr': lr, 't': 0,
                'm': [torch.zeros_like(v) for v in self.variables()],
                'v': [torch.zeros_like(v) for v in self.variables()]}

    def minimize(self, count_layers, num_steps, opt, MC_size=1):
        """tfp.math.minimize: tape -> loss -> grads -> Keras Adam -> constraints."""
        trace = []
        vs = self.variables()
        for _ in range(num_steps):
            loss = self.get_loss(count_layers, None, MC_size)
            grads = torch.autograd.grad(loss, vs)
            trace.append(float(loss.detach()))
            opt['t'] += 1
            t = opt['t']
            alpha = opt['lr'] * math.sqrt(1 - ADAM_B2 ** t) / (1 - ADAM_B1 ** t)
            with torch.no_grad():
                for v, g, m, s in zip(vs, grads, opt['m'], opt['v']):
                    m.add_((g - m) * (1 - ADAM_B1))
                    s.add_((g * g - s) * (1 - ADAM_B2))
                    v.sub_((m * alpha) / (torch.sqrt(s) + ADAM_EPS))
                    if v is self.Z_loc or v is self.intercept:   # ref:69,81
                        v.clamp_(-9, 9)
        return np.asarray(trace, np.float32)

    def fit(self, count_layers, Xc=None, min_iter=1000, max_iter=5000, add_iter=500,
            epsilon_conv=1e-2, MC_size=1, n_loss_gene=500):
        """ref:214-273."""
        self.Xc = None if Xc is None else torch.as_tensor(np.asarray(Xc), dtype=self.dtype)
        count_layers = [torch.as_tensor(np.asarray(c), dtype=self.dtype) for c in count_layers]
        for i in range(6):
            opt = self.new_adam(LEARNING_RATES[i])
            losses = self.minimize(count_layers, int(min_iter / 6), opt, MC_size)
        n_iter = min_iter + 0
        d1 = int(min(50, add_iter / 2))
        d2 = d1 * 2
        while (losses[-d2:-d1].mean() - losses[-d1:].mean() > epsilon_conv and n_iter < max_iter):
            n_iter += add_iter
            losses = np.concatenate([losses, self.minimize(count_layers, add_iter, opt, MC_size)])
        with torch.no_grad():
            lg = self.get_loss(count_layers, 0).numpy().copy()
            for _ in range(n_loss_gene - 1):
                lg += self.get_loss(count_layers, 0).numpy()
        self.loss_gene = lg / n_loss_gene
        self.losses = losses
        self.n_iter = n_iter
        return losses


def time_reference_shape(Nc, n_genes_total, counts_fn, Xc, n_batches, n_steps, MC_size=1,
                         batch_size=500000, threads=None, warmup_steps=5):
    """Time the reference-equivalent eager CPU path on a bounded sample.

    Splits genes into reference-sized batches (`ceil(batch_size/Nc)` genes,
    model_wrap.py:242) and runs `n_steps` Adam steps on `n_batches` of them.
    `counts_fn(g0, g1)` returns the L count layers for genes [g0, g1).
    Returns element-iterations per second (cells x genes x steps / s).
    """
    import time
    if threads:
        torch.set_num_threads(int(threads))
    n_gene = int(math.ceil(batch_size / float(Nc)))
    done, elapsed = 0, 0.0
    for b in range(n_batches):
        g0 = b * n_gene
        g1 = min(g0 + n_gene, n_genes_total)
        if g0 >= g1:
            break
        layers = [torch.as_tensor(np.asarray(c), dtype=torch.float32) for c in counts_fn(g0, g1)]
        Kc = 0 if Xc is None else Xc.shape[1]
        mdl = TorchBRIE2(Nc, g1 - g0, Kc, noise='torch', seed=b)
        mdl.Xc = None if Xc is None else torch.as_tensor(np.asarray(Xc), dtype=torch.float32)
        opt = mdl.new_adam(LEARNING_RATES[0])
        mdl.minimize(layers, warmup_steps, opt, MC_size)
        t0 = time.perf_counter()
        mdl.minimize(layers, n_steps, opt, MC_size)
        elapsed += time.perf_counter() - t0
        done += Nc * (g1 - g0) * n_steps
    return done / elapsed, elapsed
