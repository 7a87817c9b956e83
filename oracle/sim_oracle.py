"""Driver of oracle/sim_oracle.c + NumPy restatement of the Psi step of the simulator.  TEST INFRASTRUCTURE ONLY.

Restates /root/reference/brie/models/simulator.py:7-75 (see sim_oracle.c for the sampling scheme)."""
import ctypes
import os
import subprocess

import numpy as np

from . import philox

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "sim_oracle.c")
LIB = os.path.join(HERE, "_build", "libsim_oracle.so")
SIM_PSI_DRAW = 0xFFFFFFFE


def build(force=False):
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", SRC, "-o", LIB, "-lm"], check=True)
    return LIB


def _lib():
    lib = ctypes.CDLL(build())
    lib.sim_oracle_binomial.restype = ctypes.c_double
    lib.sim_oracle_binomial.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_uint32, ctypes.c_uint32,
                                        ctypes.c_uint64]
    lib.sim_oracle_counts.restype = None
    lib.sim_oracle_counts.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_uint64] + \
        [ctypes.c_void_p] * 6
    return lib


def binomial(n, p, gene=0, cell=0, seed=0):
    return _lib().sim_oracle_binomial(float(n), float(p), int(gene), int(cell), int(seed))


def simulate_counts(psi, total, effLen=None, seed=0, gene_offset=0):
    """(c1, c2, c3) ~ Multinomial(floor(total), phi) per element (simulator.py:45-69); c3 is None without effLen."""
    psi = np.ascontiguousarray(psi, np.float32)
    total = np.ascontiguousarray(total, np.float32)
    Nc, Ng = psi.shape
    eff = None if effLen is None else np.ascontiguousarray(effLen, np.float32)
    out = [np.zeros((Nc, Ng), np.float32) for _ in range(3)]
    ptr = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
    _lib().sim_oracle_counts(Nc, Ng, int(gene_offset), int(seed), ptr(psi), ptr(total), ptr(eff),
                             ptr(out[0]), ptr(out[1]), ptr(out[2]))
    return out[0], out[1], (out[2] if eff is not None else None)


def simulate_psi(mean_logit, sigma, seed=0, gene_offset=0):
    """Psi = expit(clip(mean + sigma_j * eps, -9, 9)) with eps from the shared stream (simulator.py:31-41)."""
    mean_logit = np.asarray(mean_logit, np.float32)
    Nc, Ng = mean_logit.shape
    eps = philox.normal(seed, SIM_PSI_DRAW, 0, Nc, Ng, gene_offset).astype(np.float32)
    z = np.clip(mean_logit + np.asarray(sigma, np.float32).reshape(1, Ng) * eps, -9.0, 9.0).astype(np.float64)
    return (1.0 / (1.0 + np.exp(-z))).astype(np.float32)
