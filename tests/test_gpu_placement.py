"""-m gpu: the placement search of the streamed arrays (brie_placement_probe / _tune / _info, include/brie_amd.h).

How fast a handle's step kernel streams depends on where the allocator put its arrays (the same
code on the same data at 8.1 or 9.5 ms per step at configs[2]).  The library times an effect-free probe kernel with
the step kernel's traffic on the handle's arrays and on up to two further sets and keeps the fastest.  What has to hold
whatever the rates are: the probe changes NOTHING, moving to another set changes nothing but addresses, and a fit that
searched is bit-identical to one that did not."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _state(sh):
    from brie_amd import _capi
    from tests import util
    st = util.device_state(sh)
    st["c1"] = sh.read(_capi.COUNT1)
    st["c2"] = sh.read(_capi.COUNT2)
    return st


def _same(a, b):
    for k in a:
        assert a[k].shape == b[k].shape and np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k


@pytest.mark.parametrize("L,storage,big", [(2, None, False), (2, None, True), (3, None, False), (2, "f32", False)])
def test_probe_and_forced_search_change_nothing_but_addresses(lib, L, storage, big):
    from brie_amd import _capi
    from tests import util
    Nc, Ng, Kc = 700, 1100, 2
    P = util.problem(Nc, Ng, Kc, L, seed=77)
    if big:                                   # some quads beyond 255: u8 / u16 tiers per gene quad
        P["counts"][0][5, 3] = 300.0
        P["counts"][1][9, 700] = 5000.0
    a = util.device_shard(P, Nc, Ng, Kc, 5, storage=storage)
    b = util.device_shard(P, Nc, Ng, Kc, 5, storage=storage)
    b.placement_tune(1, 0.0)                  # one "set" = the arrays as they are: b never moves
    ta = [a.step(4, 0.01, 1)]
    tb = [b.step(4, 0.01, 1)]
    before = _state(a)
    addr = [a.debug_address(w) for w in (_capi.Z_LOC, _capi.Z_STD_LOG, 20, 21, 22, 23)]
    rate = a.placement_probe(2)
    assert rate > 0.0
    _same(before, _state(a))                  # the probe has no effect
    info = a.placement_tune(3, 1e30)          # unreachable target: all three sets are tried, the fastest is kept
    assert info["tries"] == 3 and 0 <= info["kept"] < 3 and len(info["GBs"]) == 3 and min(info["GBs"]) > 0.0
    assert info["GBs"][info["kept"]] == max(info["GBs"]) and info["seconds"] > 0.0
    assert info["status"] == "best_of_all" and "no set reached the stop rate" in info["note"] and info["peak_extra_bytes"] > 0, info
    assert b.placement_info()["status"] == "good" and b.placement_info()["peak_extra_bytes"] == 0
    after = [a.debug_address(w) for w in (_capi.Z_LOC, _capi.Z_STD_LOG, 20, 21, 22, 23)]
    assert (after != addr) == (info["kept"] != 0)
    _same(before, _state(a))                  # ... and neither has the move
    ta.append(a.step(5, 0.005, 3))
    tb.append(b.step(5, 0.005, 3))
    assert np.array_equal(np.concatenate(ta), np.concatenate(tb))
    _same(_state(a), _state(b))
    # per-batch convergence on the moved arrays: freezing packs the active gene quads to the front (the arrays of a state
    # slab keep their place, the others swap roles with a scratch buffer), un-freezing restores the order
    mask = np.random.default_rng(3).random(Ng) < 0.4
    for sh in (a, b):
        sh.set_gene_mask(mask)
    ta.append(a.step(3, 0.01, 1))
    tb.append(b.step(3, 0.01, 1))
    for sh in (a, b):
        sh.set_gene_mask(None)
    ta.append(a.step(2, 0.01, 1))
    tb.append(b.step(2, 0.01, 1))
    assert np.array_equal(np.concatenate(ta), np.concatenate(tb))
    _same(_state(a), _state(b))
    lg_a, lg_b = a.loss_gene(3), b.loss_gene(3)
    assert np.array_equal(lg_a, lg_b)
    # b never moved.  With u8 / u16 tiers per gene quad the packing unified the tiers into FRESH count arrays: what a search
    # measured no longer describes them and the handle forgot it (the next step of a large handle searches again)
    assert b.placement_info()["tries"] == (0 if big else 1)
    a.close()
    b.close()


def test_the_first_step_of_a_large_handle_searches_by_itself(lib):
    """>= 256 MiB per step: brie_step probes before its first launch (and tries further sets while the rate is below
    the library's idea of fast).  The result is the one of a handle that was told not to search."""
    from brie_amd import _capi
    from tests import util
    Nc, Ng, Kc = 24000, 260, 1                # 6.2 M elements x 50 B = 312 MB per step
    P = util.problem(Nc, Ng, Kc, 2, seed=78)
    a = util.device_shard(P, Nc, Ng, Kc, 6)
    b = util.device_shard(P, Nc, Ng, Kc, 6)
    assert a.placement_info()["tries"] == 0
    b.placement_tune(1, 0.0)
    ta, tb = a.step(3, 0.01, 1), b.step(3, 0.01, 1)
    info = a.placement_info()
    assert 1 <= info["tries"] <= _capi.PLACEMENT_MAX_SETS and info["GBs"][info["kept"]] == max(info["GBs"])
    assert info["status"] in ("good", "best_of_all", "stopped_memory", "stopped_time"), info
    assert (info["status"] == "good") == ("note" not in info), info          # short of a fast set: the handle says why
    assert np.array_equal(ta, tb)
    assert np.array_equal(a.read(_capi.PSI), b.read(_capi.PSI))
    n = a.placement_info()["tries"]
    a.step(2, 0.01, 1)
    assert a.placement_info()["tries"] == n   # once per handle
    a.close()
    b.close()


@pytest.mark.parametrize("inject,status,tries", [(1, "stopped_error", 0), (2, "stopped_memory", 1), (3, "stopped_error", 1),
                                                  (4, "stopped_error", 1)])
def test_a_failure_inside_the_search_never_fails_the_step(lib, inject, status, tries):
    """the search is an optimisation.  With a failure injected into its first probe / a candidate's allocation /
    a candidate's probe / a candidate's copy, the first step of a large handle still runs, on the original arrays, with the
    result of a handle that never searched, and brie_placement_status says what happened."""
    from brie_amd import _capi
    from tests import util
    Nc, Ng, Kc = 24000, 260, 1
    P = util.problem(Nc, Ng, Kc, 2, seed=78)
    a = util.device_shard(P, Nc, Ng, Kc, 6)
    b = util.device_shard(P, Nc, Ng, Kc, 6)
    b.placement_tune(1, 0.0)
    a.inject_placement_failure(inject)              # an explicit call on this handle (nothing in the environment)
    ta = a.placement_tune(4, 1e30) and a.step(3, 0.01, 1)
    tb = b.step(3, 0.01, 1)
    info = a.placement_info()
    assert info["status"] == status and info["tries"] == tries and info["kept"] == 0 and info.get("note"), info
    assert np.array_equal(ta, tb)
    assert np.array_equal(a.read(_capi.PSI), b.read(_capi.PSI))
    a.close()
    b.close()


def test_replaced_count_arrays_drop_the_measurements(lib):
    """the rates of a search describe the arrays it probed.  When the count layers are re-uploaded (expanded to
    fresh fp32 layers, compacted again at the next step) the handle forgets them and the next step searches again."""
    from brie_amd import _capi
    from tests import util
    Nc, Ng, Kc = 24000, 260, 1
    P = util.problem(Nc, Ng, Kc, 2, seed=78)
    a = util.device_shard(P, Nc, Ng, Kc, 6)
    a.step(2, 0.01, 1)
    assert a.placement_info()["tries"] >= 1
    a.set_count_storage(1)                     # back to fp32 layers: fresh arrays
    info = a.placement_info()
    assert info["tries"] == 0 and info["status"] == "not_run", info
    a.step(1, 0.01, 1)
    assert a.placement_info()["tries"] >= 1
    a.close()


def test_layout_probe_times_given_layouts_inside_one_slab(lib):
    """brie_probe_layouts (the experiment aid behind DESIGN 4.3: the same memory, the arrays carved out at given offsets):
    one rate per layout, offsets checked against the slab."""
    from brie_amd import _capi
    Nc, Ng = 4000, 1024
    ld = 1024
    mat, cnt = Nc * ld * 4, Nc * ld
    packed = [i * mat for i in range(6)] + [6 * mat, 6 * mat + cnt]
    apart = [i * (mat + (8 << 20)) for i in range(6)] + [6 * (mat + (8 << 20)), 7 * (mat + (8 << 20))]
    slab = 8 * (mat + (8 << 20))
    g = _capi.probe_layouts(Nc, Ng, slab, [packed, apart], iters=2)
    assert g.shape == (2,) and (g > 0).all()
    with pytest.raises((ValueError, _capi.BrieError)):
        _capi.probe_layouts(Nc, Ng, slab, [[slab] * 8])                 # outside the slab
    with pytest.raises((ValueError, _capi.BrieError)):
        _capi.probe_layouts(Nc, Ng, slab, [[8] * 8])                    # not 16-byte aligned
