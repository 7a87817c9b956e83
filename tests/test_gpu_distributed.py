"""-m gpu: the gene-sharded COUPLED protocol (brie_step_begin -> all-reduce of per-cell statistics ->
brie_step_end) with two ranks.  Only one GPU is available on the test box, so both ranks drive handles on
cuda:0 and exchange through gloo (NCCL refuses two ranks on one device); on an 8-GPU node the same code
path runs over RCCL (GeneComm.allreduce_inplace)."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

NC, NG, KC, STEPS = 120, 520, 1, 6


def _problem(KG):
    from tests import util
    P = util.problem(NC, NG, KC, 2, seed=51)
    P["Xg"] = np.random.default_rng(8).standard_normal((NG, KG)).astype(np.float32)
    return P


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir, KG):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from brie_amd import _capi
    from brie_amd.sharding import GeneComm, gene_shard
    from tests import util
    comm = GeneComm()
    P = _problem(KG)
    g0, g1 = gene_shard(NG, rank, world)
    Ps = dict(P, counts=[c[:, g0:g1].copy() for c in P["counts"]], Xg=P["Xg"][g0:g1].copy())
    sh = _capi.Shard(NC, g1 - g0, KC, n_layers=2, seed=61, gene_offset=g0, Kg=KG, intercept_mode=1, sharded=True)
    for l in range(2):
        sh.upload(_capi.COUNT1 + l, Ps["counts"][l])
    sh.add_pseudo_count(0.01)
    sh.upload(_capi.XC, P["Xc"])
    sh.upload(_capi.XG, Ps["Xg"])
    sh.init_state()
    with pytest.raises(_capi.BrieError):
        sh.step(1, 0.01)                       # a coupled shard must use the begin/end protocol
    stat = torch.zeros(sh.rowstat_size(), dtype=torch.float32, device="cuda:0")
    trace = sh.step_sharded(STEPS, 0.01, 1, comm.allreduce_inplace, stat)
    trace = comm.allreduce_sum(trace)
    st = util.device_state(sh)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), trace=trace, g=np.array([g0, g1]), **st)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("KG", [2, 6, 70])      # registers (Kg <= 4) / Xg tile in LDS / 64-feature panels
def test_coupled_gene_shards_world2_match_single_fit(lib, tmp_path, KG):
    from tests import util
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), KG), nprocs=2, join=True)
    P = _problem(KG)
    full = util.device_shard(P, NC, NG, KC, 61, Kg=KG, mode="cell")
    tr = full.step(STEPS, 0.01, 1)
    ref = util.device_state(full)
    r = [np.load(tmp_path / ("rank%d.npz" % k)) for k in range(2)]
    np.testing.assert_allclose(r[0]["trace"], tr, rtol=1e-5)
    np.testing.assert_array_equal(r[0]["trace"], r[1]["trace"])
    for key in ("Wg_loc", "intercept", "sigma_log"):              # replicated per-cell parameters
        np.testing.assert_array_equal(r[0][key], r[1][key])
        np.testing.assert_allclose(r[0][key], ref[key], atol=2e-5)
    for key in ("Z_loc", "Z_std_log", "Wc_loc"):                  # gene-sharded state
        got = np.concatenate([r[0][key], r[1][key]], axis=1)
        d = np.abs(got - ref[key])
        assert np.percentile(d, 99.9) < 2e-5 and d.max() < 1e-3, key


FIT = dict(min_iter=120, max_iter=120, n_loss_gene=5, verbose=False, seed=3)
FNC, FNG, FKC = 60, 44, 2


def _fit_worker(rank, world, port, out_dir, ng=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import brie_amd
    from brie_amd.sharding import GeneComm
    from oracle.synth import make_problem
    from tests.fakes import FakeAnnData
    P = make_problem(FNC, ng or FNG, Kc=FKC, L=3, seed=21)
    ad = FakeAnnData({'isoform1': P["counts"][0], 'isoform2': P["counts"][1], 'ambiguous': P["counts"][2]},
                     effLen=P["effLen"])
    res = brie_amd.fitBRIE(ad, Xc=P["Xc"], LRT_index=[1], comm=GeneComm(), **FIT)
    np.savez(os.path.join(out_dir, "fit%d.npz" % rank), sigma=res.sigma, intercept=res.intercept,
             cell_coeff=res.cell_coeff, loss_gene=res.loss_gene, ELBO_gain=res.ELBO_gain, pval=res.pval,
             losses=res.losses, Psi_full=ad.layers.get('Psi', np.zeros(0)), Psi_shard=ad.layers['Psi_shard'],
             gene_range=np.array(res.gene_range))
    dist.barrier()
    dist.destroy_process_group()


def test_fitBRIE_gene_sharded_world2_real_engine(lib, tmp_path):
    """The whole gene-sharded fitBRIE (BASELINE configs[3] in miniature) with the HIP engine: two ranks, each with its
    gene block on cuda:0, collectives over gloo.  Genes are independent, so every per-gene result must equal the
    single-process fit bit for bit; only the summed loss trace is formed in another order."""
    import brie_amd
    from oracle.synth import make_problem
    from tests.fakes import FakeAnnData
    mp.spawn(_fit_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    P = make_problem(FNC, FNG, Kc=FKC, L=3, seed=21)
    ad = FakeAnnData({'isoform1': P["counts"][0], 'isoform2': P["counts"][1], 'ambiguous': P["counts"][2]},
                     effLen=P["effLen"])
    ref = brie_amd.fitBRIE(ad, Xc=P["Xc"], LRT_index=[1], **FIT)
    r = [np.load(tmp_path / ("fit%d.npz" % k)) for k in range(2)]
    assert tuple(r[0]["gene_range"]) == (0, 24) and tuple(r[1]["gene_range"]) == (24, 44)
    for key in ("sigma", "intercept", "cell_coeff", "loss_gene", "ELBO_gain", "pval"):
        np.testing.assert_array_equal(r[0][key], r[1][key])
        np.testing.assert_array_equal(r[0][key], getattr(ref, key), err_msg=key)
    np.testing.assert_allclose(r[0]["losses"], ref.losses, rtol=1e-6)
    np.testing.assert_array_equal(r[0]["Psi_full"], ref.Psi)
    np.testing.assert_array_equal(np.concatenate([r[0]["Psi_shard"], r[1]["Psi_shard"]], axis=1), ref.Psi)
    assert r[1]["Psi_full"].size == 0


# (on the CPU oracle this schedule extends once -- windowed drop 1552 -- and stops -- drop 210 < 300: trace length 40)
CFIT = dict(min_iter=120, max_iter=240, add_iter=20, epsilon_conv=300.0, n_loss_gene=3, verbose=False, seed=4, LRT_index=[])


def _coupled_fit_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import brie_amd
    from brie_amd.sharding import GeneComm
    from oracle.synth import make_problem
    from tests.fakes import FakeAnnData
    P = make_problem(FNC, FNG, Kc=1, L=2, seed=23, depth=6.0)
    Xg = np.random.default_rng(2).standard_normal((FNG, 2)).astype(np.float32)
    ad = FakeAnnData({'isoform1': P["counts"][0], 'isoform2': P["counts"][1]})
    # emulate_batches=True is ignored by a coupled model (it cannot be cut into independent batches): it stays ONE
    # sharded fit with a per-step all-reduce and must take its stopping decisions on the loss summed over the ranks
    res = brie_amd.fitBRIE(ad, Xc=P["Xc"], Xg=Xg, comm=GeneComm(), emulate_batches=True, **CFIT)
    np.savez(os.path.join(out_dir, "cfit%d.npz" % rank), losses=res.losses, sigma=res.sigma, gene_coeff=res.gene_coeff,
             loss_gene=res.loss_gene, cell_coeff=res.cell_coeff)
    dist.barrier()
    dist.destroy_process_group()


def test_coupled_sharded_fitBRIE_with_emulate_batches_stops_like_the_single_process_fit(lib, tmp_path):
    """with emulate_batches=True a coupled sharded fit used to decide its convergence extensions on
    the rank-LOCAL loss -- the ranks could extend a different number of rounds and their per-step all-reduces no longer
    paired up (a hang).  Two ranks, Kg = 2, a schedule in which extension rounds do happen: both ranks and the
    single-process fit take the same number of rounds and agree on the results."""
    import brie_amd
    from oracle.synth import make_problem
    from tests.fakes import FakeAnnData
    mp.spawn(_coupled_fit_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    P = make_problem(FNC, FNG, Kc=1, L=2, seed=23, depth=6.0)
    Xg = np.random.default_rng(2).standard_normal((FNG, 2)).astype(np.float32)
    ad = FakeAnnData({'isoform1': P["counts"][0], 'isoform2': P["counts"][1]})
    ref = brie_amd.fitBRIE(ad, Xc=P["Xc"], Xg=Xg, **CFIT)
    r = [np.load(tmp_path / ("cfit%d.npz" % k)) for k in range(2)]
    assert len(r[0]["losses"]) == len(r[1]["losses"]) == len(ref.losses)
    assert 20 < len(ref.losses) < 140                                   # extended beyond the last stage, stopped before max_iter
    np.testing.assert_allclose(r[0]["losses"], r[1]["losses"], rtol=1e-6)       # the summed trace both ranks decided on
    np.testing.assert_allclose(r[0]["losses"], ref.losses, rtol=1e-4)
    for key in ("sigma", "gene_coeff", "loss_gene", "cell_coeff"):
        np.testing.assert_array_equal(r[0][key], r[1][key])                       # gathered / replicated: same on both ranks
        np.testing.assert_allclose(r[0][key], getattr(ref, key), rtol=2e-3, atol=2e-3, err_msg=key)


# ---- the target world of EIGHT ranks on one device (round 5) --------------------------------------------------
def test_coupled_gene_shards_world8_match_single_fit(lib, tmp_path):
    """The per-step exchange of a gene-sharded COUPLED fit with eight ranks (gene features in an LDS tile, per-cell intercepts):
    eight shards of 68 / 44 genes on cuda:0, the per-cell statistics all-reduced over gloo every step -- against the unsharded
    fit.  (520 genes over 8 ranks: 7 x 68 + 44.)"""
    from tests import util
    KG = 6
    mp.spawn(_worker, args=(8, _free_port(), str(tmp_path), KG), nprocs=8, join=True)
    P = _problem(KG)
    full = util.device_shard(P, NC, NG, KC, 61, Kg=KG, mode="cell")
    tr = full.step(STEPS, 0.01, 1)
    ref = util.device_state(full)
    r = [np.load(tmp_path / ("rank%d.npz" % k)) for k in range(8)]
    assert [tuple(q["g"]) for q in r] == [(68 * k, min(68 * (k + 1), NG)) for k in range(8)]
    np.testing.assert_allclose(r[0]["trace"], tr, rtol=1e-5)
    for q in r[1:]:
        np.testing.assert_array_equal(r[0]["trace"], q["trace"])
        for key in ("Wg_loc", "intercept", "sigma_log"):          # replicated per-cell parameters: identical on every rank
            np.testing.assert_array_equal(r[0][key], q[key])
    for key in ("Wg_loc", "intercept", "sigma_log"):
        np.testing.assert_allclose(r[0][key], ref[key], atol=2e-5)
    for key in ("Z_loc", "Z_std_log", "Wc_loc"):                  # gene-sharded state, in rank order
        got = np.concatenate([q[key] for q in r], axis=1)
        d = np.abs(got - ref[key])
        assert np.percentile(d, 99.9) < 2e-5 and d.max() < 1e-3, key


def test_fitBRIE_gene_sharded_world8_real_engine(lib, tmp_path):
    """fitBRIE with the HIP engine over eight ranks on cuda:0 (120 genes: 7 x 16 + 8; LRT, effLen): every per-gene result
    equals the single-process fit bit for bit, the gathered layer on rank 0 is in rank order."""
    import brie_amd
    from oracle.synth import make_problem
    from tests.fakes import FakeAnnData
    ng = 120
    mp.spawn(_fit_worker, args=(8, _free_port(), str(tmp_path), ng), nprocs=8, join=True)
    P = make_problem(FNC, ng, Kc=FKC, L=3, seed=21)
    ad = FakeAnnData({'isoform1': P["counts"][0], 'isoform2': P["counts"][1], 'ambiguous': P["counts"][2]},
                     effLen=P["effLen"])
    ref = brie_amd.fitBRIE(ad, Xc=P["Xc"], LRT_index=[1], **FIT)
    r = [np.load(tmp_path / ("fit%d.npz" % k)) for k in range(8)]
    bounds = [tuple(int(x) for x in q["gene_range"]) for q in r]
    assert bounds == [(16 * k, min(16 * (k + 1), ng)) for k in range(8)]
    assert all(bounds[k][1] == bounds[k + 1][0] for k in range(7))
    for key in ("sigma", "intercept", "cell_coeff", "loss_gene", "ELBO_gain", "pval"):
        for q in r[1:]:
            np.testing.assert_array_equal(r[0][key], q[key])
        np.testing.assert_array_equal(r[0][key], getattr(ref, key), err_msg=key)
    np.testing.assert_allclose(r[0]["losses"], ref.losses, rtol=1e-6)
    np.testing.assert_array_equal(r[0]["Psi_full"], ref.Psi)
    np.testing.assert_array_equal(np.concatenate([q["Psi_shard"] for q in r], axis=1), ref.Psi)
