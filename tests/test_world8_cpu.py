"""CPU, gloo, world_size EIGHT: the target world of BASELINE configs[3] / configs[4].

No 8-GPU node has been available to the build, every other multi-rank test runs two ranks.  What eight ranks change is
arithmetic and ordering, and that runs here without a GPU:
  * the shard arithmetic at the configs' own sizes -- Ng 20 000 / 30 000 genes, batch_size 500 000 over 50 000 / 100 000
    cells = 10 / 5 genes per convergence batch (model_wrap.py:242), boundaries on lcm(4, genes per batch);
  * fitBRIE gene-sharded over 8 ranks with the engine's REAL control flow (tests/fakes.py::engine_on_oracle): every rank
    non-empty, the end-of-fit gather in rank order = the reference's concatenation order (model_wrap.py:241-260),
    per-batch convergence identical to the single-process fit (the eight-way all-reduce of the batch windows);
  * `python bench.py --gpus 8` starting its own eight ranks, each reporting the gene shard it would take.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from oracle.synth import make_problem

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORLD = 8
# name: Ng, cells and batch_size such that ceil(batch_size / Nc) = the config's genes per convergence batch
CASES = {"configs2_arith": dict(Ng=20000, Nc=6, batch_size=60, per_batch=10, Kc=1),     # 500 000 / 50 000 = 10
         "configs4_arith": dict(Ng=30000, Nc=6, batch_size=30, per_batch=5, Kc=2)}      # 500 000 / 100 000 = 5
FIT = dict(min_iter=60, max_iter=100, add_iter=10, epsilon_conv=0.05, n_loss_gene=2, verbose=False)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("Ng,Nc", [(20000, 50000), (30000, 100000), (5000, 10000), (500, 200)])
@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_shard_arithmetic_at_the_config_sizes(Ng, Nc, world):
    """The ranges fitBRIE hands to the ranks (brie_amd/models/wrap.py: align = lcm(4, genes per batch), quads if a rank
    would be empty): contiguous, in rank order, covering [0, Ng), none empty, no convergence batch cut in two."""
    from brie_amd.sharding import gene_shard
    per_batch = int(np.ceil(500000 / Nc))
    align = int(np.lcm(4, per_batch))
    if gene_shard(Ng, world - 1, world, align)[0] >= Ng:
        align = 4
    r = [gene_shard(Ng, k, world, align) for k in range(world)]
    assert r[0][0] == 0 and r[-1][1] == Ng
    for (a0, a1), (b0, b1) in zip(r[:-1], r[1:]):
        assert a1 == b0
    assert all(b > a for a, b in r), r
    assert all(a % align == 0 for a, _ in r)
    if align % per_batch == 0:
        assert all(a % per_batch == 0 for a, _ in r)               # every batch lives on one rank
    if (Ng, world) == (20000, 8):
        assert r == [(2500 * k, 2500 * (k + 1)) for k in range(8)]
    if (Ng, world) == (30000, 8):                                  # 3750 is no multiple of lcm(4, 5) = 20: 7 x 3760 + 3680
        assert r == [(3760 * k, min(3760 * (k + 1), 30000)) for k in range(8)]


def _worker(rank, world, port, out_dir, case):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      OMP_NUM_THREADS="1")
    import torch
    torch.set_num_threads(1)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import brie_amd.models.wrap as wrap
    from brie_amd.sharding import GeneComm
    from tests.fakes import FakeAnnData, engine_on_oracle
    wrap.BRIE2 = E = engine_on_oracle()
    c = CASES[case]
    P = make_problem(c["Nc"], c["Ng"], Kc=c["Kc"], L=2, seed=21, depth=6.0)
    ad = FakeAnnData({'isoform1': P["counts"][0], 'isoform2': P["counts"][1]})
    res = wrap.fitBRIE(ad, Xc=P["Xc"], comm=GeneComm(), batch_size=c["batch_size"], **FIT)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), sigma=res.sigma, intercept=res.intercept, cell_coeff=res.cell_coeff,
             loss_gene=res.loss_gene, Psi=res.Psi, gene_range=np.array(res.gene_range), n_losses=len(res.losses),
             n_iter_batch=np.asarray(E.instances[0].n_iter_batch), has_full=rank == 0 and 'Psi' in ad.layers,
             Psi_full=ad.layers['Psi'] if (rank == 0 and 'Psi' in ad.layers) else np.zeros(0))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", sorted(CASES))
def test_fitBRIE_over_eight_ranks_equals_the_single_process_fit(tmp_path, case):
    c = CASES[case]
    mp.spawn(_worker, args=(WORLD, _free_port(), str(tmp_path), case), nprocs=WORLD, join=True)
    r = [np.load(tmp_path / ("rank%d.npz" % k)) for k in range(WORLD)]
    import brie_amd.models.wrap as wrap
    from tests.fakes import FakeAnnData, engine_on_oracle
    saved = wrap.BRIE2
    wrap.BRIE2 = E = engine_on_oracle()
    try:
        P = make_problem(c["Nc"], c["Ng"], Kc=c["Kc"], L=2, seed=21, depth=6.0)
        ad = FakeAnnData({'isoform1': P["counts"][0], 'isoform2': P["counts"][1]})
        ref = wrap.fitBRIE(ad, Xc=P["Xc"], batch_size=c["batch_size"], **FIT)
    finally:
        wrap.BRIE2 = saved
    # ranges: rank order, contiguous, non-empty, whole batches
    bounds = [tuple(int(x) for x in q["gene_range"]) for q in r]
    assert bounds[0][0] == 0 and bounds[-1][1] == c["Ng"] and all(b > a for a, b in bounds)
    assert all(bounds[k][1] == bounds[k + 1][0] for k in range(WORLD - 1))
    assert all(a % c["per_batch"] == 0 for a, _ in bounds)
    # every rank holds the same gathered per-gene vectors, in the reference's concatenation order
    for key in ("sigma", "intercept", "cell_coeff", "loss_gene"):
        for q in r[1:]:
            np.testing.assert_array_equal(r[0][key], q[key])
        np.testing.assert_array_equal(r[0][key], getattr(ref, key))
    np.testing.assert_array_equal(np.concatenate([q["Psi"] for q in r], axis=1), ref.Psi)
    assert bool(r[0]["has_full"])
    np.testing.assert_array_equal(r[0]["Psi_full"], ref.Psi)           # rank 0 gathered the layer in rank order
    # per-batch convergence: the batches stop where they stop in the single-process fit, and they do stop apart
    n_iter = np.concatenate([q["n_iter_batch"] for q in r])
    np.testing.assert_array_equal(n_iter, E.instances[0].n_iter_batch)
    assert len(n_iter) == c["Ng"] // c["per_batch"] and len(set(n_iter.tolist())) > 1
    assert all(int(q["n_losses"]) == len(ref.losses) for q in r)


@pytest.mark.parametrize("config,Ng", [("c3", 20000), ("c5", 30000)])
def test_bench_starts_eight_ranks_and_each_takes_its_shard(config, Ng):
    """`python bench.py --gpus 8` (no launcher in the environment) -> eight ranks over 127.0.0.1; with
    BRIE_BENCH_ECHO_RANK=shards the ranks rendezvous over gloo and report the gene range each would fit instead of
    touching a GPU: the ranges the strong-scaling line of SCALE_rNN.json is made of."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["BRIE_BENCH_ECHO_RANK"] = "shards"
    argv = ["--gpus", "8", "--steps", "5", "--warmup", "1", "--config", config]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    echo = json.loads(lines[0])
    assert echo["world"] == 8 and echo["argv"] == argv and echo["master_addr"] == "127.0.0.1" and echo["local_world"] == "8"
    shards = echo["shards"]
    assert [s["rank"] for s in shards] == list(range(8)) and [s["local_rank"] for s in shards] == list(range(8))
    assert shards[0]["genes"][0] == 0 and shards[-1]["genes"][1] == Ng
    assert all(a["genes"][1] == b["genes"][0] for a, b in zip(shards[:-1], shards[1:]))
    assert all(s["genes"][1] > s["genes"][0] and s["genes"][0] % 4 == 0 for s in shards)
