"""The boundary from plain C: tests/c_abi/smoke.c (one handle) and tests/c_abi/shard_client.c (N ranks, one process per GPU:
RCCL communicator through the C ABI, gene shards, end-of-fit all-gather) are compiled with gcc against include/brie_amd.h and
linked to libbrie_amd.so (CPU: build + link only; `-m gpu`: run them).  No Python, no torch in those processes."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "c_abi", "_build", "smoke")


def _build(name="smoke"):
    from brie_amd.build import compile_library
    compile_library()
    exe = os.path.join(os.path.dirname(EXE), name)
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    lib_dir = os.path.join(ROOT, "brie_amd", "lib")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", os.path.join(ROOT, "tests", "c_abi", name + ".c"),
                    "-I" + os.path.join(ROOT, "include"), "-L" + lib_dir, "-lbrie_amd", "-lm",
                    "-Wl,-rpath," + lib_dir, "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe], check=True)
    return exe


def test_c_client_compiles_and_links_against_the_header():
    exe = _build()
    out = subprocess.run(["nm", "-u", exe], check=True, capture_output=True, text=True).stdout
    for sym in ("brie_create", "brie_upload", "brie_step", "brie_read", "brie_destroy", "brie_last_error"):
        assert sym in out                                  # resolved from libbrie_amd.so at run time


@pytest.mark.gpu
def test_c_client_runs_on_the_gpu():
    exe = _build()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "loss" in r.stdout and "mc_size" in r.stdout    # the refused call left its message in brie_last_error()
    print(r.stdout.strip())


def test_multi_rank_c_client_compiles_and_links_against_the_header():
    exe = _build("shard_client")
    out = subprocess.run(["nm", "-u", exe], check=True, capture_output=True, text=True).stdout
    for sym in ("brie_comm_available", "brie_comm_unique_id", "brie_comm_init", "brie_comm_allgather", "brie_comm_allreduce",
                "brie_comm_destroy", "brie_create", "brie_loss_gene"):
        assert sym in out
    r = subprocess.run([exe], capture_output=True, text=True)      # no arguments: usage, before anything touches a GPU
    assert r.returncode == 2 and "usage" in r.stderr


def _gpus():
    import torch
    return torch.cuda.device_count()                               # counting devices does not initialise the GPU


@pytest.mark.gpu
@pytest.mark.parametrize("world", ["one", "node"])
def test_multi_rank_c_client_gathers_what_one_handle_fits(world):
    """N fresh processes (started by a shell script before any of them touches the GPU), RCCL unique id through a file,
    brie_comm_init -> sharded fits -> brie_comm_allgather; rank 0 re-fits all genes on one handle and demands the gathered
    [Wc_loc, intercept, sigma, loss_gene] bit for bit.  World 1 on any box; the node's world when it has more than one GPU
    (RCCL refuses two ranks on one device)."""
    n = 1 if world == "one" else _gpus()
    if world == "node" and n < 2:
        pytest.skip("needs >= 2 GPUs (this node has %d)" % n)
    _build("shard_client")
    r = subprocess.run(["bash", os.path.join(ROOT, "tests", "c_abi", "run_shards.sh"), str(n), str(max(1, _gpus()))],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert ("OK world=%d" % n) in r.stdout and "mismatches=0" in r.stdout
    print(r.stdout.strip())
