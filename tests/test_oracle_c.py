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


def test_c_oracle_is_clean_under_address_and_ub_sanitizers(tmp_path):
    """The checker itself under ASan + UBSan on the CPU (GPU sanitizers are unavailable on this pool): both
    precisions, 2- and 3-layer likelihood, ragged shapes; any finding aborts with a non-zero exit code."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for extra in ([], ["-DBRIE_ORACLE_F64"]):
        exe = str(tmp_path / ("oracle_san" + ("64" if extra else "32")))
        subprocess.run(["gcc", "-O1", "-g", "-fopenmp", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                        "-fno-omit-frame-pointer"] + extra +
                       [os.path.join(root, "tests", "c_abi", "oracle_sanitize.c"), os.path.join(root, "oracle", "brie_oracle.c"),
                        "-lm", "-o", exe], check=True)
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", OMP_NUM_THREADS="3")
        r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "clean" in r.stdout
