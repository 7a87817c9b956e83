"""Property tests (hypothesis) of the host-side arithmetic every rank must agree on: gene sharding, the cuts of a
gene range into sequential parts, id matching, Benjamini-Hochberg.  CPU only."""
import numpy as np
from hypothesis import given, settings, strategies as st

from brie_amd.models.wrap import _part_bounds
from brie_amd.preprocessing import match
from brie_amd.sharding import gene_shard
from brie_amd.stats import fdr_bh


@settings(max_examples=300, deadline=None)
@given(Ng=st.integers(1, 50000), world=st.integers(1, 16), k=st.integers(1, 64))
def test_gene_shards_partition_the_genes(Ng, world, k):
    align = 4 * k
    edges = [gene_shard(Ng, r, world, align) for r in range(world)]
    assert edges[0][0] == 0 and edges[-1][1] == Ng
    for (a0, a1), (b0, b1) in zip(edges, edges[1:]):
        assert a1 == b0 and a0 <= a1
    for a0, a1 in edges:
        assert a0 % align == 0 or a0 == Ng                 # every shard starts on a whole batch / quad
    sizes = [b - a for a, b in edges]
    assert sum(sizes) == Ng and max(sizes) - min(s for s in sizes if s) < 2 * align + Ng // world + 1


@settings(max_examples=300, deadline=None)
@given(lo=st.integers(0, 1000), n=st.integers(1, 40000), n_parts=st.integers(1, 12), sb=st.integers(1, 80),
       nb=st.integers(1, 50))
def test_part_bounds_cover_the_range_with_nonempty_pieces(lo, n, n_parts, sb, nb):
    lo *= 4
    super_batch = sb * 256
    parts = _part_bounds(lo, lo + n, n_parts, super_batch, nb)
    assert parts[0][0] == lo and parts[-1][1] == lo + n
    assert all(a < b for a, b in parts) and all(p[1] == q[0] for p, q in zip(parts, parts[1:]))
    assert len(parts) <= max(n_parts, -(-n // super_batch))
    if len(parts) == n_parts and n_parts > 1:               # inner boundaries sit on gene quads at least
        assert all((a - lo) % 4 == 0 for a, _ in parts)


@settings(max_examples=200, deadline=None)
@given(ref=st.lists(st.integers(0, 30), max_size=40), new=st.lists(st.integers(0, 30), max_size=40, unique=True))
def test_match_agrees_with_its_definition(ref, new):
    got = match(ref, new)
    seen = set()
    for key, j in zip(ref, got):
        if key in new and key not in seen:
            assert new[j] == key                           # first occurrence: the index of the equal new id
        else:
            assert j is None                               # absent, or a repeated reference id (uniq_ref_only)
        seen.add(key)
    every = match(ref, new, uniq_ref_only=False)
    assert all((j is None) == (key not in new) and (j is None or new[j] == key) for key, j in zip(ref, every))


@settings(max_examples=200, deadline=None)
@given(st.lists(st.floats(0, 1, allow_nan=False), min_size=1, max_size=200))
def test_bh_adjusted_values_are_monotone_and_bounded(p):
    p = np.asarray(p)
    q = fdr_bh(p)
    assert np.all(q >= p - 1e-15) and np.all(q <= 1 + 1e-15)
    order = np.argsort(p, kind="stable")
    assert np.all(np.diff(q[order]) >= -1e-12)             # adjusted p-values keep the order of the raw ones
