"""CPU: `python bench.py --gpus N` must start its own ranks.

The driver launches the 1-GPU bench as `python bench.py --gpus 1 ...`; the same command shape with N > 1 and no
launcher in the environment used to exit at once.  Now the parent -- before importing torch.cuda or calling HIP --
starts `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>` as a child process and returns
its exit code.  BRIE_BENCH_ECHO_RANK=1 makes the ranks report what reached them instead of touching a GPU.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_launcher_command_relays_every_argument():
    sys.path.insert(0, ROOT)
    import bench
    argv = ["--gpus", "4", "--steps", "7", "--warmup", "2", "--config", "c2", "--no-pmc"]
    cmd = bench.launcher_command(argv, 4, 29517)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29517"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == argv                                       # unchanged, after the script
    # a rank's single-process children (the rocprofv3 --pmc runs) must not inherit the launcher's variables
    os.environ["RANK"], os.environ["WORLD_SIZE"] = "0", "4"
    try:
        env = bench.standalone_env()
    finally:
        del os.environ["RANK"], os.environ["WORLD_SIZE"]
    assert "RANK" not in env and "WORLD_SIZE" not in env and "PATH" in env


def test_bench_starts_its_own_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["BRIE_BENCH_ECHO_RANK"] = "1"
    argv = ["--gpus", "2", "--steps", "5", "--warmup", "1", "--config", "c2"]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout                                 # ONE line, from rank 0
    echo = json.loads(lines[0])
    assert echo["world"] == 2 and echo["argv"] == argv and echo["master_addr"] == "127.0.0.1"


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", BRIE_BENCH_ECHO_RANK="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=2" in (p.stderr + p.stdout)


def test_native_communicator_leg_is_bounded_and_reported():
    """The C-ABI communicator's gather has never run between two GPUs on the build's boxes: bench.py runs it LAST under a
    watchdog.  A leg that hangs is reported (native_error) and the caller is told not to touch the process group again;
    a leg that raises is reported too; a leg that works records its time and the comparison with the torch gather."""
    import threading
    import numpy as np
    sys.path.insert(0, ROOT)
    import bench

    class Native(object):
        def allgather(self, buf):
            return np.stack([buf, buf + 1.0])

    class Comm(object):
        backend = "nccl"

        def __init__(self, mode):
            self.mode, self.gate = mode, threading.Event()

        def native_comm(self, local_rank):
            if self.mode == "hang":
                self.gate.wait()                       # never set: RCCL set-up that does not come back
            if self.mode == "raise":
                raise RuntimeError("ncclCommInitRank failed")
            return None if self.mode == "absent" else Native()

    local = np.arange(6, dtype=np.float32).reshape(2, 3)
    ranges = [(0, 3), (3, 6)]
    full = np.concatenate([local, local + 1.0], axis=1)
    for mode in ("ok", "absent", "raise", "hang"):
        info = {}
        back = bench.native_allgather_leg(info, (Comm(mode), local, ranges, full), 0, 2, timeout_s=0.5)
        if mode == "ok":
            assert back and info["native_equals_torch"] is True and info["allgather_native_ms"] >= 0
        elif mode == "absent":
            assert back and "no native communicator" in info["native"]
        elif mode == "raise":
            assert back and "ncclCommInitRank failed" in info["native_error"]
        else:
            assert not back and "did not return" in info["native_error"]
    os.environ["BRIE_BENCH_NATIVE_COMM"] = "0"
    try:
        info = {}
        assert bench.native_allgather_leg(info, (Comm("hang"), local, ranges, full), 0, 2, timeout_s=0.5)
        assert "skipped" in info["native"]
    finally:
        del os.environ["BRIE_BENCH_NATIVE_COMM"]
