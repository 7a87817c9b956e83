"""The reference-derived pin of the optimiser loop -- ONE command away, wherever TensorFlow is.

    python tests/golden/make_golden_tf.py          # needs tensorflow + tensorflow_probability AND /root/reference

Build container only (never shipped to the GPU box; writes DATA, no reference source): loads the reference's
brie/models/model_TFProb.py by path, gives its BRIE2 the initial state of the shared Philox stream through `init_obj`
(model_TFProb.py:45,62-65), replaces the unseeded `tfd.Normal(Z_loc, Z_std).sample(MC_size)` (model_TFProb.py:159) by
loc + scale * eps with eps from oracle/philox.py (one draw id per loss evaluation, advanced by a tf.Variable so that it
also advances inside tfp.math.minimize's traced loop), runs the reference's own `fit` on the two tiny golden problems of
make_golden.py and stores losses, loss_gene and the fitted state in tests/golden/ref_tf_traj_<case>.npz.
tests/test_oracle.py::test_oracle_against_the_tensorflow_reference compares oracle/brie_oracle.py with those files when they
exist and reports "parity unpinned (TensorFlow absent)" otherwise.  TensorFlow / TFP are in neither interpreter nor
wheelhouse of the round-1..5 images, so this script has NOT run: it is the recipe, written against the TF 2.15 / TFP 0.23
API the reference documents (doc/install.rst:76-77).
"""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/brie/models/model_TFProb.py"
CASES = {"lik2_kc2": dict(Nc=48, Ng=36, Kc=2, L=2, MC=1), "eff3_kc1_mc3": dict(Nc=40, Ng=28, Kc=1, L=3, MC=3)}
SEED, MIN_ITER = 1234, 120


def load_reference():
    spec = importlib.util.spec_from_file_location("ref_model_TFProb", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)                  # imports tensorflow, tensorflow_probability (ImportError without them)
    return mod


def pinned_model(ref, c, P, seed):
    import tensorflow as tf
    from tensorflow_probability import distributions as tfd
    from oracle import philox
    from oracle.brie_oracle import OracleBRIE2
    Nc, Ng, Kc = c["Nc"], c["Ng"], c["Kc"]
    o = OracleBRIE2(Nc, Ng, Kc, effLen=P["effLen"], seed=seed, dtype=np.float32)     # Philox init = Model_init's fields
    init = type("Init", (), {})()
    init.intercept, init.sigma = tf.constant(o.intercept), tf.constant(np.exp(o.sigma_log))
    init.Z_loc, init.Z_std = tf.constant(o.Z_loc), tf.constant(np.exp(o.Z_std_log))
    init.Wc_loc, init.Wg_loc = tf.constant(o.Wc_loc), tf.zeros([Nc, 0])
    draw = tf.Variable(0, dtype=tf.int64, trainable=False)

    def eps_of(d, n):                             # (n, Nc, Ng) float32 of draw id d: k = 0..n-1
        return np.stack([philox.normal(seed, int(d), k, Nc, Ng) for k in range(int(n))]).astype(np.float32)

    class Pinned(ref.BRIE2):
        @property
        def Z(self):                              # a genuine tfd.Normal (kl_divergence sees what it always saw) ...
            q = tfd.Normal(self.Z_loc, self.Z_std)

            def sample(n, *a, **kw):              # ... whose sample() is the reparameterised draw on the shared stream
                d = draw.assign_add(1) - 1
                e = tf.numpy_function(lambda dd: eps_of(dd, n), [d], tf.float32)
                e.set_shape((int(n), Nc, Ng))
                return q.loc + q.scale * e
            q.sample = sample
            return q

    m = Pinned(Nc, Ng, Kc=Kc, effLen=None if P["effLen"] is None else np.asarray(P["effLen"], np.float32), init_obj=init)
    return m, draw


def main():
    try:
        ref = load_reference()
    except ImportError as exc:
        raise SystemExit("make_golden_tf.py needs tensorflow and tensorflow_probability (%s): parity stays unpinned" % exc)
    import tensorflow as tf
    import tensorflow_probability as tfp
    from oracle.brie_oracle import add_pseudo_count
    from oracle.synth import make_problem
    for name, c in CASES.items():
        P = make_problem(c["Nc"], c["Ng"], Kc=c["Kc"], L=c["L"], seed=77, theta=2.0)      # the problems of make_golden.py
        cnt = [np.asarray(x, np.float32) for x in add_pseudo_count(P["counts"])]           # model_wrap.py:113-117
        m, draw = pinned_model(ref, c, P, SEED)
        init = {"Z_loc": m.Z_loc.numpy(), "Z_std_log": m.Z_std_log.numpy(), "Wc_loc": m.Wc_loc.numpy(),
                "intercept": m.intercept.numpy(), "sigma_log": m.sigma_log.numpy()}
        # max_iter = min_iter: the six stages only (the extension rule is host logic, tests/test_host_logic.py)
        losses = m.fit(cnt, Xc=np.asarray(P["Xc"], np.float32), min_iter=MIN_ITER, max_iter=MIN_ITER, verbose=False,
                       MC_size=c["MC"])
        np.savez_compressed(
            os.path.join(HERE, "ref_tf_traj_%s.npz" % name), losses=losses.numpy(), loss_gene=m.loss_gene.numpy(),
            draws=int(draw.numpy()), seed=SEED, min_iter=MIN_ITER, MC=c["MC"], tf=tf.__version__, tfp=tfp.__version__,
            Z_loc=m.Z_loc.numpy(), Z_std_log=m.Z_std_log.numpy(), Wc_loc=m.Wc_loc.numpy(), intercept=m.intercept.numpy(),
            sigma_log=m.sigma_log.numpy(), Psi=m.Psi.numpy(), Psi95CI=m.Psi95CI, **{"init_" + k: v for k, v in init.items()})
        print("wrote ref_tf_traj_%s.npz: %d losses, %d noise draws" % (name, len(losses), int(draw.numpy())))


if __name__ == "__main__":
    main()
