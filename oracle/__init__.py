"""CPU oracle for the brie-quant ELBO hot path.  TEST INFRASTRUCTURE ONLY.

Nothing in the product package (`brie_amd/`) may import, call, link or execute
anything in this directory.  The only permitted users are `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`, and there
only as the checker / the reported baseline, never as the thing measured.

What it restates
----------------
The per-gene variational ELBO optimisation of BRIE2 (reference v2.3.0), i.e.
`/root/reference/brie/models/model_TFProb.py:12-273` plus the host logic of
`/root/reference/brie/models/model_wrap.py:88-260`.  The arithmetic of that
path lives in third-party dependencies that are NOT under /root/reference and
are NOT installed here:

    tensorflow              (requirements.txt:18  ">=2.0.0";  docs-tested 2.15.1)
    tensorflow-probability  (requirements.txt:19  ">=0.8.0"; docs-tested 0.23.0)

so their published algorithms are restated (Normal reparameterised sampling,
Normal-Normal KL with expm1, log_sigmoid, reduce_logsumexp, Keras Adam with
epsilon outside the bias correction, variable constraints applied after the
update, tfp.math.minimize tracing the pre-update loss).

PARITY UNPINNED by the reference's own tests
--------------------------------------------
The reference ships no tests, golden vectors or fixtures for this path
(`tests/README.rst:1-3`, `.travis.yml:6-9`) and its implementation cannot be
imported or run here (TF/TFP absent), so there is no reference output to pin
the restatement against.  What pins it instead (see tests/test_oracle_*.py):

 1. the two reference fragments that DO import by file path in the build
    container -- `brie/models/base_model.py:get_CI95` (29-36) and
    `BRIE_base_lik` (20-27) -- evaluated on random grids by
    `tests/golden/make_golden.py` and committed as `tests/golden/*.npz`;
 2. torch autograd of a line-by-line restatement of `get_loss`
    (`brie_oracle_torch.py`) against the hand-derived gradients used here;
 3. analytic identities (KL = 0 and zero gradient at q = prior; zero counts
    => posterior -> prior; Philox4x32-10 Random123 known-answer vectors);
 4. recovery of simulated ground truth (generative recipe of
    `brie/models/simulator.py:22-69`).

Modules
-------
philox.py             shared counter-based N(0,1) stream (Philox4x32-10 + Box-Muller)
brie_oracle.py        NumPy restatement with hand-derived gradients (fp32 / fp64)
brie_oracle_torch.py  eager torch-CPU autograd restatement in the reference's
                      execution shape; also the `cpu_baseline` ("port") of bench.py
brie_oracle.c         fused C / OpenMP restatement of the step (second implementation; "cpu_baseline_fused"); built twice:
                      fp32 = the reference's precision, -DBRIE_ORACLE_F64 = the same code in double (the
                      precision-independent answer of profiles/psi_delta.py and the parity rule in tests/util.py)
c_oracle.py           gcc build + ctypes driver of brie_oracle.c
sim_oracle.c / .py    count simulator (brie/models/simulator.py): exact binomial / multinomial sampling from the
                      Philox stream; distribution pinned against scipy (tests/test_oracle_sim.py)
host_stats.py         chi2 / Benjamini-Hochberg restatement for the LRT driver
synth.py              seeded synthetic count generator (SURVEY.md 8d recipe)
"""
