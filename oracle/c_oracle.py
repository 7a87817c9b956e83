"""ctypes driver of oracle/brie_oracle.c (fused C / OpenMP restatement).  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "brie_oracle.c")
LIB = os.path.join(HERE, "_build", "libbrie_oracle.so")
LIB_F64 = os.path.join(HERE, "_build", "libbrie_oracle_f64.so")
LIB_B = os.path.join(HERE, "_build", "libbrie_oracle_b.so")
LIB_MUT = os.path.join(HERE, "_build", "libbrie_oracle_mut.so")
# negative controls (brie_oracle.c, enum MUT_*): deliberately wrong variants the parity rules must reject
MUTANTS = {"none": 0, "adam_eps_torch": 1, "adam_eps_1e8": 2, "no_clip": 3, "beta2_double": 4, "kl_no_expm1": 6,
           "mc_same_noise": 9, "no_bias_corr": 10, "lik_grad_1pct": 11, "lik_grad_01pct": 12, "kl_grad_1pct": 13,
           "sigma_grad_sign": 14}


def _cpu_has_fma():
    try:
        with open("/proc/cpuinfo") as fh:
            return " fma " in fh.read().replace("\n", " ")
    except OSError:
        return False


def build(force=False, f64=False, variant_b=False, mutants=False):
    """gcc -O3 -fopenmp -> oracle/_build/libbrie_oracle.so (fp32, the reference's precision) or, with f64=True,
    libbrie_oracle_f64.so (the same code with every quantity in double).  -ffp-contract=off: no fused
    multiply-adds, so the fp32 build rounds after every operation like the eager reference does.
    variant_b=True: libbrie_oracle_b.so, "o32b" -- a second fp32 evaluation (float Box-Muller, reversed cell order with
    fp32 partial sums, -ffp-contract=fast and FMA instructions where the host has them; see the header of brie_oracle.c).
    mutants=True: libbrie_oracle_mut.so -- the o32b build plus brie_oracle_set_mutant (negative controls; a mutant is "some
    other fp32 evaluation" with ONE deliberate error, the position a wrong HIP kernel would be in)."""
    assert not (f64 and (variant_b or mutants))
    variant_b = variant_b or mutants
    lib = LIB_MUT if mutants else LIB_B if variant_b else (LIB_F64 if f64 else LIB)
    if force or not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(SRC):
        os.makedirs(os.path.dirname(lib), exist_ok=True)
        flags = ["-DBRIE_ORACLE_F64"] if f64 else []
        if variant_b:
            flags = ["-DBRIE_ORACLE_B", "-ffp-contract=fast"] + (["-mfma"] if _cpu_has_fma() else [])
            if mutants:
                flags.append("-DBRIE_ORACLE_MUTANTS")
        else:
            flags = ["-ffp-contract=off"] + flags
        subprocess.run(["gcc", "-O3"] + flags + ["-fopenmp", "-shared", "-fPIC", SRC, "-o", lib, "-lm"], check=True)
    return lib


class _Problem(ctypes.Structure):
    _fields_ = [("Nc", ctypes.c_int32), ("Ng", ctypes.c_int32), ("Kc", ctypes.c_int32), ("n_layers", ctypes.c_int32),
                ("has_efflen", ctypes.c_int32), ("mc", ctypes.c_int32), ("train_b", ctypes.c_int32),
                ("train_lam", ctypes.c_int32), ("gene_offset", ctypes.c_int64), ("seed", ctypes.c_uint64)]


class COracle(object):
    """State + optimiser slots as float32 arrays, stepped by the C kernel; mirrors OracleBRIE2's fields."""

    def __init__(self, counts, Xc, effLen=None, seed=0, gene_offset=0, intercept=None, sigma=None, init=None,
                 dtype=np.float32, variant_b=False, mutant=None):
        from .brie_oracle import OracleBRIE2
        self.dtype = np.dtype(dtype)
        self.lib = ctypes.CDLL(build(f64=self.dtype == np.float64, variant_b=variant_b, mutants=mutant is not None))
        if mutant is not None:             # a global of the loaded library, like the o32b knobs
            variant_b = True
            if self.lib.brie_oracle_set_mutant(ctypes.c_int(MUTANTS[mutant])) != 0:
                raise ValueError("not the mutant build")
        assert self.lib.brie_oracle_real_bytes() == self.dtype.itemsize
        self.lib.brie_oracle_set_parts(ctypes.c_int(0))
        if variant_b:                      # the knobs are globals of the loaded library: every o32b run starts from the defaults
            self.lib.brie_oracle_b_config(ctypes.c_int(1), ctypes.c_int(1), ctypes.c_int(64))
        dt = self.dtype
        self.counts = [np.ascontiguousarray(c, dt) for c in counts]
        self.Nc, self.Ng = self.counts[0].shape
        self.Xc = np.ascontiguousarray(Xc if Xc is not None else np.zeros((self.Nc, 0)), dt)
        self.Kc = self.Xc.shape[1]
        self.effLen = None if effLen is None else np.ascontiguousarray(effLen, dt)
        self.seed, self.gene_offset, self.draw, self.t = int(seed), int(gene_offset), 0, 0
        self.train_b, self.train_lam = intercept is None, sigma is None
        o = OracleBRIE2(self.Nc, self.Ng, self.Kc, effLen=effLen, intercept=intercept, sigma=sigma, seed=seed,
                        gene_offset=gene_offset, dtype=np.float32, init=init)
        f = lambda a: np.ascontiguousarray(a, dt)
        self.Z_loc, self.Z_std_log, self.Wc_loc = f(o.Z_loc), f(o.Z_std_log), f(o.Wc_loc)
        self.intercept, self.sigma_log = f(o.intercept).reshape(-1), f(o.sigma_log).reshape(-1)
        self.reset_optimizer()

    def reset_optimizer(self):
        self.t = 0
        self.slots = {k: (np.zeros_like(getattr(self, k)), np.zeros_like(getattr(self, k)))
                      for k in ("Z_loc", "Z_std_log", "Wc_loc", "intercept", "sigma_log")}

    def minimize(self, n_steps, lr, MC_size=1):
        p = _Problem(self.Nc, self.Ng, self.Kc, len(self.counts), int(self.effLen is not None), int(MC_size),
                     int(self.train_b), int(self.train_lam), self.gene_offset, self.seed)
        trace = np.zeros(n_steps, self.dtype)
        ptr = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
        c3 = self.counts[2] if len(self.counts) > 2 else None
        s = self.slots
        rc = self.lib.brie_oracle_steps(
            ctypes.byref(p), ctypes.c_int32(n_steps), ctypes.c_double(float(np.float32(lr))), ctypes.c_int32(self.t),
            ctypes.c_uint32(self.draw), ptr(self.counts[0]), ptr(self.counts[1]), ptr(c3), ptr(self.Xc), ptr(self.effLen),
            ptr(self.Z_loc), ptr(self.Z_std_log), ptr(s["Z_loc"][0]), ptr(s["Z_loc"][1]), ptr(s["Z_std_log"][0]),
            ptr(s["Z_std_log"][1]), ptr(self.Wc_loc), ptr(s["Wc_loc"][0]), ptr(s["Wc_loc"][1]), ptr(self.intercept),
            ptr(s["intercept"][0]), ptr(s["intercept"][1]), ptr(self.sigma_log), ptr(s["sigma_log"][0]),
            ptr(s["sigma_log"][1]), ptr(trace))
        if rc != 0:
            raise MemoryError("brie_oracle_steps")
        self.t += n_steps
        self.draw += n_steps
        return trace

    @property
    def Psi(self):
        from .brie_oracle import sigmoid
        return sigmoid(self.Z_loc)

    def threads(self):
        return int(self.lib.brie_oracle_threads())

    def b_config(self, float_noise=1, reverse=1, chunk=64):
        """Knobs of the o32b build (a member of the null ensemble, tests/golden/psi_ensemble_manifest.json): Box-Muller in
        float or exact, the thread's cells in reverse or forward order, cells per fp32 partial sum."""
        if self.lib.brie_oracle_b_config(ctypes.c_int(int(float_noise)), ctypes.c_int(int(reverse)), ctypes.c_int(int(chunk))) != 0:
            raise ValueError("b_config: not the o32b build (variant_b=True)")

    def set_parts(self, n):
        """Cut the cells into n parts (own accumulators, summed in part order) whatever the thread count: the arithmetic
        of a run on n OpenMP threads, reproducible on any host.  0 = one part per thread (the default)."""
        self.lib.brie_oracle_set_parts(ctypes.c_int(int(n)))

    def set_threads(self, n):
        """OpenMP threads of the fused pass (torch.distributed.run exports OMP_NUM_THREADS=1 to its ranks)."""
        self.lib.brie_oracle_set_threads(ctypes.c_int(int(n)))
        return self.threads()
