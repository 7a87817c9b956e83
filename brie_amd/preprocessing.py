"""Gene filter and id matching used by the brie-quant front end
(/root/reference/brie/utils/preprocessing.py:5-83, /root/reference/brie/utils/base_utils.py:5-59)."""
import numpy as np


def _colsum(m):
    return np.asarray(m.sum(0)).reshape(-1)


def filter_genes(data, min_counts=0, min_cells=0, min_counts_uniq=0, min_cells_uniq=0, min_MIF_uniq=0.001,
                 uniq_layers=['isoform1', 'isoform2'], ambg_layers=['ambiguous'], copy=False):
    """Keep genes with enough total / unique counts, enough expressing cells and a minor isoform
    frequency >= min_MIF_uniq among unique counts; adds `n_counts`, `n_counts_uniq` to `.var`.
    Returns the filtered object (a new one: sub-setting is never in place here)."""
    uniq = [data.layers[k] for k in uniq_layers]
    unique_counts = uniq[0] + uniq[1] if len(uniq) > 1 else uniq[0]
    for extra in uniq[2:]:
        unique_counts = unique_counts + extra
    total_counts = unique_counts
    for k in ambg_layers:
        total_counts = total_counts + data.layers[k]
    n_tot, n_unq = _colsum(total_counts), _colsum(unique_counts)
    keep = np.ones(data.shape[1], dtype=bool)
    keep &= n_tot >= min_counts
    keep &= _colsum(total_counts > 0) >= min_cells
    keep &= n_unq >= min_counts_uniq
    keep &= _colsum(unique_counts > 0) >= min_cells_uniq
    keep &= _colsum(uniq[0]) >= min_MIF_uniq * n_unq
    keep &= _colsum(uniq[1]) >= min_MIF_uniq * n_unq
    out = data[:, keep]
    out.var['n_counts'] = n_tot[keep]
    out.var['n_counts_uniq'] = n_unq[keep]
    dropped = int(np.sum(~keep))
    if dropped:
        rules = [("%d cells with any count", min_cells), ("%d total counts", min_counts),
                 ("%d cells with unique counts", min_cells_uniq), ("%d unique counts", min_counts_uniq),
                 ("%.4f minor isoform frequency", min_MIF_uniq)]
        print("Filtered out %d genes with less than " % dropped + " or ".join(t % v for t, v in rules if v > 0))
    return out


def match(ref_ids, new_ids):
    """Index into `new_ids` for every entry of `ref_ids` (None where absent); `new_ids` unique."""
    lookup = {}
    for j, key in enumerate(new_ids):
        lookup.setdefault(key, j)
    return np.array([lookup.get(key) for key in ref_ids], dtype=object)
