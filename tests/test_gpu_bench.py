"""-m gpu: `bench.py` beyond one rank.  The driver measures N = 1 on its 1-GPU boxes; the N > 1 line
-- gene shards, per-GPU roofline entries, the end-of-fit all-gather of the per-gene vectors (BASELINE configs[3]: "RCCL
weight all-gather"; replaces the `concate` of model_wrap.py:260) and its bit-for-bit check against rank 0's own re-fit --
must not be exercised by hand only.  Two ranks share GPU 0 here (gloo: RCCL refuses two ranks on one device); the RCCL
branch and the library's own communicator run with a world of one rank."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--config", "c1", "--steps", "5", "--warmup", "2", "--no-pmc", "--cpu-seconds", "1"]


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", BRIE_BENCH_STRICT="1", **extra)
    return env


def _line(p):
    if p.returncode != 0:              # the whole story, not pytest's abbreviated repr of it
        pytest.fail("bench.py exited with %d\n--- stdout (tail)\n%s\n--- stderr (tail)\n%s"
                    % (p.returncode, p.stdout[-1500:], p.stderr[-6000:]), pytrace=False)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_two_ranks_give_one_complete_line(lib):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + ARGS,
                       env=_env(BRIE_BENCH_SINGLE_DEVICE="1"), capture_output=True, text=True, timeout=600, cwd=ROOT)
    d = _line(p)
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["value"] > 0 and d["scaling"] == "strong"
    assert d["warmup"] == 2 and d["preconditioning"]["probe_launches"] > 0       # the W + K steps as asked; the device kept busy before
    per = d["roofline"]["per_gpu"]
    assert [g["rank"] for g in per] == [0, 1] and sum(g["genes"] for g in per) == d["config"]["Ng"]
    assert all(g["avg_kernel_ms"] > 0 and 0 < g["frac"] < 1 for g in per)
    assert d["roofline"]["frac"] == min(g["frac"] for g in per)
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1 and "cpu_baseline_all_cores" in d
    ga = d["allgather"]
    assert ga["backend"] == "gloo" and ga["recomputed_on_rank0"]["bit_identical"] is True
    assert ga["recomputed_on_rank0"]["first_gene_of_quads"] == [per[0]["genes"]]
    assert "holds" in d["psi_delta_vs_cpu_ref"]["rule"]


def test_eight_ranks_on_one_gpu_give_one_complete_line(lib):
    """The world the driver's scaling run ends at (BASELINE configs[3]: eight gene shards), on ONE device over gloo: the
    launcher, the eight shards of configs[1] (5 000 genes -> 8 x 628 / 604), barrier + maximum over ranks, the gather of
    every rank's per-gene vectors in rank order, rank 0's bit-for-bit re-fit of a gene quad of each of the seven other shards."""
    args = ["--config", "c2", "--steps", "5", "--warmup", "2", "--no-pmc", "--no-e2e", "--no-f32-leg", "--no-cpu-baseline", "--no-psi-check"]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"] + args,
                       env=_env(BRIE_BENCH_SINGLE_DEVICE="1"), capture_output=True, text=True, timeout=900, cwd=ROOT)
    d = _line(p)
    per = d["roofline"]["per_gpu"]
    assert d["n_gpus"] == 8 and [g["rank"] for g in per] == list(range(8)) and sum(g["genes"] for g in per) == d["config"]["Ng"] == 5000
    assert all(g["genes"] > 0 and g["genes"] % 4 == 0 for g in per)
    ga = d["allgather"]
    assert ga["backend"] == "gloo" and ga["recomputed_on_rank0"]["bit_identical"] is True
    starts = [sum(g["genes"] for g in per[:k]) for k in range(1, 8)]
    assert ga["recomputed_on_rank0"]["first_gene_of_quads"] == starts and len(ga["ms_per_step_per_rank"]) == 8
    assert d["ms_per_step"] == pytest.approx(max(ga["ms_per_step_per_rank"]), rel=1e-6)       # MAX over ranks


def test_one_rank_on_rccl_runs_the_gather_through_both_communicators(lib):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = _env(BRIE_BENCH_FORCE_GATHER="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--no-psi-check",
                        "--no-cpu-baseline", "--no-e2e", "--no-f32-leg"] + ARGS[:6] + ["--no-pmc"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    d = _line(p)
    ga = d["allgather"]
    assert d["n_gpus"] == 1 and ga["backend"] == "nccl" and ga["rccl_ranks"] == 1
    assert ga.get("native_equals_torch") is True and ga["allgather_native_ms"] >= 0, ga
    assert d["roofline"]["rccl_ranks"] == 1
