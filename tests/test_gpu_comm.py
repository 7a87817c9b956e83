"""-m gpu: the RCCL entry points of the C ABI (brie_comm_*, brie_attach_comm; include/brie_amd.h).

A 1-GPU box can only form a communicator of ONE rank -- RCCL refuses two ranks on one device -- but that already
runs every call through librccl on the hardware (init, all-gather, all-reduce on the handle's stream inside
brie_step).  The two-rank test needs two GPUs and skips cleanly otherwise; the two-rank protocol itself is covered
on one GPU over gloo in tests/test_gpu_distributed.py and on CPU in tests/test_distributed_cpu.py.
"""
import os
import socket

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu


def test_single_rank_communicator_collectives(lib):
    import torch
    from brie_amd import _capi
    uid = _capi.Comm.unique_id()
    assert len(uid) == _capi.COMM_ID_BYTES and any(uid)
    c = _capi.Comm(0, 0, 1, uid)
    x = np.arange(1000, dtype=np.float32) * 0.5
    np.testing.assert_array_equal(c.allgather(x), x[None, :])
    np.testing.assert_array_equal(c.allreduce(x), x)
    np.testing.assert_array_equal(c.allreduce(x.astype(np.float64), "max"), x.astype(np.float64))
    t = torch.arange(4096, dtype=torch.float32, device="cuda:0")
    c.allreduce_device(t.data_ptr(), t.numel())
    np.testing.assert_array_equal(t.cpu().numpy(), np.arange(4096, dtype=np.float32))
    with pytest.raises(ValueError):
        _capi.Comm(0, 3, 2, uid)
    c.close()


@pytest.mark.parametrize("Kg,mode", [(2, 'gene'), (6, 'gene'), (0, 'cell'), (3, 'cell'), (70, 'cell')])
def test_in_library_allreduce_equals_unsharded_coupled_fit(lib, Kg, mode):
    """A handle created as ONE gene shard of a coupled fit (sharded=1) with a communicator attached runs
    local sums -> RCCL all-reduce -> Adam inside brie_step; with a world of one rank that is exactly the
    unsharded coupled fit."""
    from brie_amd import _capi
    Nc, Ng, Kc = 90, 260, 1
    P = util.problem(Nc, Ng, Kc, 2)
    rng = np.random.default_rng(5)
    P["Xg"] = rng.normal(size=(Ng, Kg)).astype(np.float32)
    ref = util.device_shard(P, Nc, Ng, Kc, 17, Kg=Kg, mode=mode)
    tr_ref = ref.step(12, 0.01, 1)

    c = _capi.Comm(0, 0, 1, _capi.Comm.unique_id())
    L = len(P["counts"])
    sh = _capi.Shard(Nc, Ng, Kc, n_layers=L, seed=17, Kg=Kg, intercept_mode=1 if mode == 'cell' else 0, sharded=True)
    for l in range(L):
        sh.upload(_capi.COUNT1 + l, P["counts"][l])
    sh.add_pseudo_count(0.01)
    sh.upload(_capi.XC, P["Xc"])
    if Kg:
        sh.upload(_capi.XG, P["Xg"])
    sh.init_state()
    with pytest.raises(_capi.BrieError):
        sh.step(1, 0.01, 1)                      # a shard of a coupled fit cannot step without the exchange
    sh.attach_comm(c)
    tr = sh.step(12, 0.01, 1)
    np.testing.assert_array_equal(tr, tr_ref)
    for k, a in util.device_state(sh).items():
        np.testing.assert_array_equal(a, util.device_state(ref)[k], err_msg=k)
    # the explicit protocol with the exchange done by the library
    _capi._check(sh.lib, sh.lib.brie_step_begin(sh._h, 0.01, 1))
    import ctypes
    loss = ctypes.c_float()
    _capi._check(sh.lib, sh.lib.brie_step_end(sh._h, ctypes.byref(loss)))
    assert loss.value == ref.step(1, 0.01, 1)[0]
    sh.attach_comm(None)
    sh.close()
    ref.close()
    c.close()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _two_rank_worker(rank, world, id_file, out_dir):
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import time
    from brie_amd import _capi
    if rank == 0:
        uid = _capi.Comm.unique_id()
        with open(id_file + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(id_file + ".tmp", id_file)
    else:
        while not os.path.exists(id_file):
            time.sleep(0.05)
        uid = open(id_file, "rb").read()
    c = _capi.Comm(rank, rank, world, uid)
    g = c.allgather(np.full(5, rank + 1.0, np.float32))
    r = c.allreduce(np.array([rank + 1.0, 10.0], np.float64))
    # coupled fit, genes split over the two ranks, exchange inside the library
    from brie_amd.sharding import gene_shard
    Nc, Ng, Kc, Kg = 70, 96, 1, 2
    P = util.problem(Nc, Ng, Kc, 2)
    P["Xg"] = np.random.default_rng(5).normal(size=(Ng, Kg)).astype(np.float32)
    g0, g1 = gene_shard(Ng, rank, world)
    sh = _capi.Shard(Nc, g1 - g0, Kc, n_layers=2, seed=17, Kg=Kg, sharded=True, device=rank, gene_offset=g0)
    for l in range(2):
        sh.upload(_capi.COUNT1 + l, P["counts"][l][:, g0:g1])
    sh.add_pseudo_count(0.01)
    sh.upload(_capi.XC, P["Xc"])
    sh.upload(_capi.XG, P["Xg"][g0:g1])
    sh.init_state()
    sh.attach_comm(c)
    tr = c.allreduce(sh.step(10, 0.01, 1).astype(np.float64))
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), gather=g, reduce=r, trace=tr, Wg=sh.read(_capi.WG_LOC),
             Z=sh.read(_capi.Z_LOC))
    sh.close()
    c.close()


def test_communicator_over_every_gpu_of_the_node(lib, tmp_path):
    """One rank per visible GPU (2, 4, 8 ... as the node has; skipped on the build's 1-GPU boxes -- RCCL refuses two ranks on
    one device): brie_comm_allgather / _allreduce between real GPUs, and a coupled fit whose genes are split over all of
    them with the per-step exchange inside the library, against the unsharded fit on one GPU."""
    import torch
    world = torch.cuda.device_count()
    if world < 2:
        pytest.skip("needs two or more GPUs: RCCL refuses two ranks on one device")
    import torch.multiprocessing as mp
    mp.spawn(_two_rank_worker, args=(world, str(tmp_path / "id.bin"), str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / ("r%d.npz" % k)) for k in range(world)]
    for k in range(world):
        np.testing.assert_array_equal(r[k]["gather"], np.array([[j + 1.0] * 5 for j in range(world)], np.float32))
        np.testing.assert_array_equal(r[k]["reduce"], [world * (world + 1) / 2.0, 10.0 * world])
    from brie_amd import _capi
    Nc, Ng, Kc, Kg = 70, 96, 1, 2
    P = util.problem(Nc, Ng, Kc, 2)
    P["Xg"] = np.random.default_rng(5).normal(size=(Ng, Kg)).astype(np.float32)
    ref = util.device_shard(P, Nc, Ng, Kc, 17, Kg=Kg)
    tr = ref.step(10, 0.01, 1)
    np.testing.assert_allclose(r[0]["trace"], tr, rtol=1e-5)
    for k in range(1, world):
        np.testing.assert_array_equal(r[0]["Wg"], r[k]["Wg"])
    np.testing.assert_allclose(r[0]["Wg"], ref.read(_capi.WG_LOC), atol=2e-5)
    np.testing.assert_allclose(np.concatenate([q["Z"] for q in r], axis=1), ref.read(_capi.Z_LOC), atol=2e-5)


def _native_gather_worker(rank, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    import brie_amd
    from brie_amd.sharding import GeneComm
    from oracle.synth import make_problem
    from tests.fakes import FakeAnnData
    P = make_problem(80, 52, Kc=1, L=2, seed=12)
    kw = dict(LRT_index=[0], min_iter=60, max_iter=60, n_loss_gene=3, verbose=False, seed=5)
    ref = brie_amd.fitBRIE(FakeAnnData({'isoform1': P["counts"][0], 'isoform2': P["counts"][1]}), Xc=P["Xc"], **kw)
    comm = GeneComm(device=torch.device("cuda", 0))
    comm.always_gather = True          # a world of one shards nothing: drive the sharded branch through RCCL anyway
    ad = FakeAnnData({'isoform1': P["counts"][0], 'isoform2': P["counts"][1]})
    res = brie_amd.fitBRIE(ad, Xc=P["Xc"], comm=comm, **kw)
    path_in_fit = comm.last_gather_path
    # the library's communicator also answers on its own, and equals the torch path
    x = np.arange(3 * 52, dtype=np.float32).reshape(3, 52)
    nat = comm.allgather_genes(x, 52, native=True)
    tor = comm.allgather_genes(x, 52, native=False)
    np.savez(os.path.join(out_dir, "native.npz"), path=path_in_fit, native_exists=comm.native_comm() is not None,
             same=all(np.array_equal(getattr(res, k), getattr(ref, k)) for k in
                      ("sigma", "intercept", "cell_coeff", "loss_gene", "ELBO_gain", "Psi", "losses")),
             gather_ok=np.array_equal(nat, x) and np.array_equal(tor, x), keys=sorted(ad.varm))
    dist.barrier()
    dist.destroy_process_group()


def test_fitBRIE_end_of_fit_gather_runs_through_the_library_communicator(lib, tmp_path):
    """with the process group on RCCL the per-gene vectors of a gene-sharded fitBRIE are gathered by
    brie_comm_allgather (librccl called from libbrie_amd.so), not by torch.distributed; replaces the `concate` of
    model_wrap.py:260.  One rank on this box: the branch, the communicator set-up over the torch store and the RCCL
    calls all execute; the result equals the unsharded fit bit for bit."""
    import torch.multiprocessing as mp
    mp.spawn(_native_gather_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    r = np.load(tmp_path / "native.npz")
    assert bool(r["native_exists"]) and str(r["path"]) == "brie_comm_allgather"    # what fitBRIE's own gather went through
    assert bool(r["same"]) and bool(r["gather_ok"])
    assert "cell_coeff" in list(r["keys"])
