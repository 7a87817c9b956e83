"""Light-weight stand-ins for the tfd distribution objects the reference's BRIE2 hands out as attributes
(`BRIE2.Z`, `.PsiDist`, `.Z_prior`; /root/reference/brie/models/model_TFProb.py:97-127).  Host-side accessors over
arrays read from the device -- the optimisation loop never touches them."""
import numpy as np
from scipy.special import expit, logit, ndtri


class Normal(object):
    """tfd.Normal(loc, scale): the members BRIE callers use (sample / log_prob / mean / stddev / quantile / kl)."""

    def __init__(self, loc, scale):
        self.loc, self.scale = np.asarray(loc, np.float32), np.asarray(scale, np.float32)

    def mean(self):
        return np.broadcast_to(self.loc, np.broadcast(self.loc, self.scale).shape)

    def stddev(self):
        return np.broadcast_to(self.scale, np.broadcast(self.loc, self.scale).shape)

    def sample(self, sample_shape=(), seed=None):
        shape = tuple(np.atleast_1d(sample_shape).astype(int)) if np.size(sample_shape) else ()
        eps = np.random.default_rng(seed).standard_normal(shape + np.broadcast(self.loc, self.scale).shape)
        return (self.loc + self.scale * eps).astype(np.float32)

    def log_prob(self, x):
        z = (np.asarray(x, np.float64) - self.loc) / self.scale
        return (-0.5 * z * z - np.log(self.scale) - 0.5 * np.log(2 * np.pi)).astype(np.float32)

    def quantile(self, q):
        return (self.loc + self.scale * ndtri(q)).astype(np.float32)

    def kl_divergence(self, other):
        """KL(self || other), TFP's _kl_normal_normal (used at model_TFProb.py:208)."""
        d = (self.loc.astype(np.float64) - other.loc) / other.scale
        r = np.log(self.scale.astype(np.float64)) - np.log(other.scale)
        return (0.5 * d * d + 0.5 * np.expm1(2 * r) - r).astype(np.float32)


class LogitNormal(object):
    """tfd.LogitNormal(loc, scale) (model_TFProb.py:97-100): sigmoid of a Normal."""

    def __init__(self, loc, scale):
        self.loc, self.scale = np.asarray(loc, np.float32), np.asarray(scale, np.float32)
        self._z = Normal(loc, scale)

    def sample(self, sample_shape=(), seed=None):
        return expit(self._z.sample(sample_shape, seed)).astype(np.float32)

    def quantile(self, q):
        return expit(self._z.quantile(q)).astype(np.float32)

    def log_prob(self, x):
        x = np.asarray(x, np.float64)
        return (self._z.log_prob(logit(x)) - np.log(x) - np.log1p(-x)).astype(np.float32)
