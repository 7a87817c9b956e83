"""CPU, world_size 2, gloo: the gene-sharded fitBRIE path (brie_amd/sharding.py).

One process per rank; each fits its contiguous gene block (with the
oracle-backed stand-in instead of the GPU engine) and the per-gene vectors are
all-gathered.  Result must equal the single-process fit gene for gene.
"""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from oracle.synth import make_problem

FIT = dict(min_iter=60, max_iter=60, n_loss_gene=3, verbose=False)
NC, NG, KC = 24, 22, 1          # 22 genes over 2 ranks -> 12 + 10 (boundary multiple of 4)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import brie_amd.models.wrap as wrap
    from brie_amd.sharding import GeneComm
    from tests.fakes import FakeAnnData, OracleBackedBRIE2
    wrap.BRIE2 = OracleBackedBRIE2
    P = make_problem(NC, NG, Kc=KC, L=2, seed=12)
    ad = FakeAnnData({'isoform1': P["counts"][0], 'isoform2': P["counts"][1]})
    comm = GeneComm()
    res = wrap.fitBRIE(ad, Xc=P["Xc"], LRT_index=[0], comm=comm, **FIT)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), sigma=res.sigma, intercept=res.intercept,
             cell_coeff=res.cell_coeff, loss_gene=res.loss_gene, ELBO_gain=res.ELBO_gain, fdr=res.fdr,
             losses=res.losses, Psi=res.Psi, gene_range=np.array(res.gene_range),
             Psi_full=ad.layers.get('Psi', np.zeros(0)), has_shard='Psi_shard' in ad.layers,
             red=comm.allreduce_sum(np.array([rank + 1.0, 2.0])))
    dist.barrier()
    dist.destroy_process_group()


def test_gene_sharded_fit_world2_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    # single-process reference run
    import brie_amd.models.wrap as wrap
    from tests.fakes import FakeAnnData, OracleBackedBRIE2
    saved = wrap.BRIE2
    wrap.BRIE2 = OracleBackedBRIE2
    try:
        P = make_problem(NC, NG, Kc=KC, L=2, seed=12)
        ad = FakeAnnData({'isoform1': P["counts"][0], 'isoform2': P["counts"][1]})
        ref = wrap.fitBRIE(ad, Xc=P["Xc"], LRT_index=[0], **FIT)
    finally:
        wrap.BRIE2 = saved
    assert tuple(r0["gene_range"]) == (0, 12) and tuple(r1["gene_range"]) == (12, 22)
    np.testing.assert_array_equal(r0["red"], [3.0, 4.0])
    for key in ("sigma", "intercept", "cell_coeff", "loss_gene", "ELBO_gain", "fdr", "losses"):
        np.testing.assert_array_equal(r0[key], r1[key])                 # every rank holds the gathered result
    assert r0["sigma"].shape == (1, NG) and r0["ELBO_gain"].shape == (NG, 1)
    np.testing.assert_allclose(r0["sigma"], ref.sigma, atol=2e-6)
    np.testing.assert_allclose(r0["cell_coeff"], ref.cell_coeff, atol=2e-6)
    np.testing.assert_allclose(r0["loss_gene"], ref.loss_gene, rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(r0["losses"], ref.losses, rtol=1e-5)
    np.testing.assert_allclose(np.concatenate([r0["Psi"], r1["Psi"]], axis=1), ref.Psi, atol=2e-6)
    # rank 0 holds the gathered layer, rank 1 only its shard
    np.testing.assert_allclose(r0["Psi_full"], ref.Psi, atol=2e-6)
    assert r1["Psi_full"].size == 0 and bool(r0["has_shard"]) and bool(r1["has_shard"])
