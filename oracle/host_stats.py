"""Host-side LRT statistics restated.  TEST INFRASTRUCTURE ONLY.

/root/reference/brie/models/model_wrap.py:189-196:
    pval = chi2.sf(2 * ELBO_gain, df=1)
    fdr[:, i] = statsmodels multipletests(pval[:, i], method="fdr_bh")[1]
statsmodels is a third-party dependency (requirements.txt, unpinned) that is
absent here; its published fdr_bh algorithm (Benjamini-Hochberg step-up with
the running minimum taken from the largest p-value down, clipped at 1) is
restated below in the plainest possible form.
"""
import numpy as np
from scipy.stats import chi2


def pval_from_gain(ELBO_gain):
    return chi2.sf(2 * np.asarray(ELBO_gain, np.float64), df=1)


def fdr_bh(pvals):
    """Plain-loop Benjamini-Hochberg adjusted p-values (multipletests(...,'fdr_bh')[1])."""
    p = np.asarray(pvals, np.float64)
    n = len(p)
    order = sorted(range(n), key=lambda i: p[i])
    adj = np.empty(n)
    running = 1.0
    for rank in range(n, 0, -1):
        i = order[rank - 1]
        running = min(running, p[i] * n / rank)
        adj[i] = min(running, 1.0)
    return adj
