"""Gene filter and id matching used by the brie-quant front end
(/root/reference/brie/utils/preprocessing.py:5-83, /root/reference/brie/utils/base_utils.py:5-59)."""
import numpy as np


def _colsum(m):
    return np.asarray(m.sum(0)).reshape(-1)


def filter_genes(data, min_counts=0, min_cells=0, min_counts_uniq=0, min_cells_uniq=0, min_MIF_uniq=0.001,
                 uniq_layers=['isoform1', 'isoform2'], ambg_layers=['ambiguous'], copy=False):
    """Keep genes with enough total / unique counts, enough expressing cells and a minor isoform
    frequency >= min_MIF_uniq among unique counts; adds `n_counts`, `n_counts_uniq` to `.var`.
    copy=False (the reference's default, preprocessing.py:37,62,83): `data` itself is sub-set in place through its
    `_inplace_subset_var(mask)` (anndata.AnnData and brie_amd.io.CountData have it) and None is returned;
    copy=True: a filtered copy is returned and `data` is left alone."""
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
    if copy:
        out = data[:, keep]
    else:
        if not hasattr(data, "_inplace_subset_var"):
            raise TypeError("filter_genes(copy=False) needs an object with _inplace_subset_var(mask) "
                            "(anndata.AnnData, brie_amd.io.CountData); pass copy=True for a filtered copy")
        data._inplace_subset_var(keep)
        out = data
    out.var['n_counts'] = n_tot[keep]
    out.var['n_counts_uniq'] = n_unq[keep]
    dropped = int(np.sum(~keep))
    if dropped:
        rules = [("%d cells with any count", min_cells), ("%d total counts", min_counts),
                 ("%d cells with unique counts", min_cells_uniq), ("%d unique counts", min_counts_uniq),
                 ("%.4f minor isoform frequency", min_MIF_uniq)]
        print("Filtered out %d genes with less than " % dropped + " or ".join(t % v for t, v in rules if v > 0))
    return out if copy else None


def match(ref_ids, new_ids, uniq_ref_only=True):
    """Index into `new_ids` for every entry of `ref_ids` (None where absent); `new_ids` unique (base_utils.py:5-59).

    The reference walks both sorted lists and, with uniq_ref_only=True (its default), moves past a new id once it
    has been matched: of several EQUAL ref ids only one gets the index, the others None (the first in order of
    appearance here; the reference's unstable argsort leaves that open).  uniq_ref_only=False: every duplicate
    gets the index."""
    lookup = {}
    for j, key in enumerate(new_ids):
        lookup.setdefault(key, j)
    out, used = [], set()
    for key in ref_ids:
        j = lookup.get(key)
        if j is not None and uniq_ref_only:
            if key in used:
                j = None
            used.add(key)
        out.append(j)
    return np.array(out, dtype=object)
