"""CPU: the fused C/OpenMP restatement (oracle/brie_oracle.c) against the NumPy oracle -- two independent
implementations of the same step must agree to fp32 rounding."""
import numpy as np
import pytest

from oracle.brie_oracle import OracleBRIE2, add_pseudo_count
from oracle.c_oracle import COracle
from oracle.synth import make_problem


@pytest.mark.parametrize("L,Kc,MC", [(2, 0, 1), (2, 3, 2), (3, 1, 3)])
def test_c_oracle_matches_numpy_oracle(L, Kc, MC):
    Nc, Ng = 70, 50
    P = make_problem(Nc, Ng, Kc=Kc, L=L, seed=91)
    cnt = add_pseudo_count(P["counts"])
    o = OracleBRIE2(Nc, Ng, Kc, effLen=P["effLen"], seed=17, dtype=np.float32, gene_offset=8)
    c = COracle(cnt, P["Xc"], effLen=P["effLen"], seed=17, gene_offset=8)
    np.testing.assert_array_equal(c.Z_loc, o.Z_loc)
    for lr in (0.01, 0.02):
        o.reset_optimizer()
        c.reset_optimizer()
        tr_o = o.minimize(cnt, P["Xc"], 6, lr, MC)
        tr_c = c.minimize(6, lr, MC)
        np.testing.assert_allclose(tr_c, tr_o, rtol=3e-5)
    for name, got in (("Z_loc", c.Z_loc), ("Z_std_log", c.Z_std_log), ("Wc_loc", c.Wc_loc),
                      ("intercept", c.intercept[None, :]), ("sigma_log", c.sigma_log[None, :])):
        ref = getattr(o, name)
        if ref.size:
            d = np.abs(got - ref)
            assert np.percentile(d, 99.9) < 2e-5 and d.max() < 1e-3, name
    assert c.draw == o.draw == 12


def test_second_fp32_build_is_the_same_algorithm_within_the_tolerances_the_device_is_granted():
    """o32b (oracle/brie_oracle.c -DBRIE_ORACLE_B; the NumPy oracle's variant_b): the null of tests/util.py::psi_null_rule.
    It must be the SAME algorithm evaluated another way: its noise within 2e-6 of the defined stream (the tolerance
    tests/test_gpu_parity.py::test_noise_stream_matches_oracle grants the device), a few steps within the short-horizon
    bounds the device is held to -- and NOT bit-identical, or it would be no null at all."""
    import ctypes
    from oracle import philox
    from oracle.c_oracle import build
    libs = {v: ctypes.CDLL(build(variant_b=v)) for v in (False, True)}
    out = {v: np.zeros((40, 25, 4), np.float32) for v in libs}
    for v, lib in libs.items():
        lib.brie_oracle_normal4.argtypes = [ctypes.c_uint32] * 4 + [ctypes.c_uint64, ctypes.c_void_p]
        for cell in range(40):
            for quad in range(25):
                lib.brie_oracle_normal4(quad, cell, 7, 2, 123456789012, out[v][cell, quad].ctypes.data_as(ctypes.c_void_p))
    ref = philox.normal(123456789012, 7, 2, 40, 100).reshape(40, 25, 4)
    np.testing.assert_array_equal(out[False], ref)                       # the first build IS the defined stream
    d = np.abs(out[True].astype(np.float64) - ref)
    assert 0 < d.max() <= 2e-6 and (out[True] != ref).mean() > 0.2
    dn = np.abs(philox.normal(5, 3, 1, 64, 64, float_box_muller=True).astype(np.float64) - philox.normal(5, 3, 1, 64, 64))
    assert 0 < dn.max() <= 2e-6
    Nc, Ng, Kc = 90, 60, 2
    P = make_problem(Nc, Ng, Kc=Kc, L=3, seed=92)
    cnt = add_pseudo_count(P["counts"])
    a = COracle(cnt, P["Xc"], effLen=P["effLen"], seed=19)
    b = COracle(cnt, P["Xc"], effLen=P["effLen"], seed=19, variant_b=True)
    na = OracleBRIE2(Nc, Ng, Kc, effLen=P["effLen"], seed=19, dtype=np.float32)
    nb = OracleBRIE2(Nc, Ng, Kc, effLen=P["effLen"], seed=19, dtype=np.float32, variant_b=True)
    np.testing.assert_allclose(b.minimize(5, 0.01, 2), a.minimize(5, 0.01, 2), rtol=3e-5)
    np.testing.assert_allclose(nb.minimize(cnt, P["Xc"], 5, 0.01, 2), na.minimize(cnt, P["Xc"], 5, 0.01, 2), rtol=3e-5)
    for x, y in ((a, b), (na, nb)):
        for name in ("Z_loc", "Z_std_log", "Wc_loc", "intercept", "sigma_log"):
            dd = np.abs(np.asarray(getattr(x, name), np.float64) - np.asarray(getattr(y, name), np.float64))
            assert np.percentile(dd, 99.9) < 2e-5 and dd.max() < 1e-3, name
        assert not np.array_equal(np.asarray(x.Z_loc), np.asarray(y.Z_loc))


def test_c_oracle_is_clean_under_address_and_ub_sanitizers(tmp_path):
    """The checker itself under ASan + UBSan on the CPU (GPU sanitizers are unavailable on this pool): both
    precisions, 2- and 3-layer likelihood, ragged shapes; any finding aborts with a non-zero exit code."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for extra in ([], ["-DBRIE_ORACLE_F64"], ["-DBRIE_ORACLE_B", "-ffp-contract=fast"]):
        exe = str(tmp_path / ("oracle_san" + "_".join(extra).replace("-", "").replace("=", "")))
        subprocess.run(["gcc", "-O1", "-g", "-fopenmp", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                        "-fno-omit-frame-pointer"] + extra +
                       [os.path.join(root, "tests", "c_abi", "oracle_sanitize.c"), os.path.join(root, "oracle", "brie_oracle.c"),
                        "-lm", "-o", exe], check=True)
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", OMP_NUM_THREADS="3")
        r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "clean" in r.stdout


def test_null_ensemble_members_are_reproducible_and_gene_slices_are_exact():
    """What the pre-registered null ensemble (tests/golden/psi_ensemble_manifest.json) relies on:
    * a member is defined by the number of PARTS the cells are cut into (own per-gene sums, added in part order), not by
      the host's cores: set_parts(n) on any thread count is bit-identical to the run on n OpenMP threads (round 4's draws);
    * the knobs of the o32b build (float / exact Box-Muller, reversed / forward order, cells per fp32 partial sum) change
      the run, and every new COracle starts from the defaults;
    * genes are independent and the noise is keyed by the global gene index: the first k genes of a run ARE the run of the
      first k genes (the fixtures hold 64-gene slices of round 4's 128-gene caches)."""
    from oracle.c_oracle import COracle
    Nc, Ng, Kc = 301, 40, 2
    P = make_problem(Nc, Ng, Kc=Kc, L=3, seed=5)
    cnt = add_pseudo_count(P["counts"])

    def run(threads, parts=0, cfg=None, genes=Ng, vb=True):
        o = COracle([np.ascontiguousarray(x[:, :genes]) for x in cnt], P["Xc"], effLen=P["effLen"][:genes], seed=3, variant_b=vb)
        o.set_threads(threads)
        if parts:
            o.set_parts(parts)
        if cfg:
            o.b_config(*cfg)
        for lr in (0.001, 0.01):
            o.reset_optimizer()
            o.minimize(60, lr, 3)
        return o.Z_loc.copy(), o.Wc_loc.copy(), o.sigma_log.copy()

    for n in (3, 7):
        a, b = run(n), run(2, parts=n)
        assert all(np.array_equal(x, y) for x, y in zip(a, b)), n
    base = run(4)
    assert not np.array_equal(base[0], run(3)[0])                     # another cut of the sums IS another evaluation
    exact = run(4, cfg=(0, 0, 128))
    assert not np.array_equal(base[0], exact[0]) and np.abs(base[0] - exact[0]).max() < 1e-3
    again = run(4)                                                      # the knobs are globals of the library: reset per instance
    assert all(np.array_equal(x, y) for x, y in zip(base, again))
    part = run(4, genes=20)
    assert np.array_equal(part[0], base[0][:, :20]) and np.array_equal(part[1], base[1][:, :20]) and np.array_equal(part[2], base[2][:20])
    o32, o32_8 = run(4, vb=False), run(2, parts=4, vb=False)            # the fp32 oracle itself: parts, not threads, define it
    assert all(np.array_equal(x, y) for x, y in zip(o32, o32_8))
