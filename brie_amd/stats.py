"""Host-side statistics of the LRT driver (model_wrap.py:189-196)."""
import numpy as np
from scipy.stats import chi2


def elbo_gain_pval(ELBO_gain):
    """`chi2.sf(2 * ELBO_gain, df=1)` (model_wrap.py:190)."""
    return chi2.sf(2 * np.asarray(ELBO_gain), df=1)


def fdr_bh(pvals):
    """Benjamini-Hochberg adjusted p-values == statsmodels
    `multipletests(p, method="fdr_bh")[1]` (model_wrap.py:193-195), vectorised."""
    p = np.asarray(pvals, dtype=np.float64).ravel()
    n = p.size
    if n == 0:
        return p.copy()
    order = np.argsort(p, kind="mergesort")
    scaled = p[order] * n / np.arange(1, n + 1)
    adj_sorted = np.minimum(np.minimum.accumulate(scaled[::-1])[::-1], 1.0)
    adj = np.empty(n)
    adj[order] = adj_sorted
    return adj
