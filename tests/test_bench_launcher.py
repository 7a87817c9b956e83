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
