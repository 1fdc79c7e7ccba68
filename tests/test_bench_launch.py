"""bench.py's launcher half, which needs no GPU: `python bench.py --gpus N` with WORLD_SIZE unset starts its own ranks
through torch.distributed.run (the old code asserted instead and could never yield a scaling point)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_self_launch_command_is_the_drivers_own():
    import bench
    cmd = bench.self_launch_command(4, ["--gpus", "4", "--steps", "7"], port=29555)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29555"
    assert cmd[-5:] == [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "7"]
    free = bench.self_launch_command(2, [])
    assert 1024 < int(free[free.index("--master-port") + 1]) < 65536


def test_gpus_2_without_a_launcher_reaches_both_ranks():
    """no GPU here: every rank must stop at bench.py's own "needs a GPU" exit -- which proves the ranks were started with
    WORLD_SIZE=2 by bench.py itself -- and the launcher's non-zero code must come back; stdout stays empty"""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("the GPU variant of this test is tests/test_bench_contract_gpu.py::test_two_ranks_without_a_launcher")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, cwd=ROOT, env=env, timeout=300)
    assert r.returncode != 0
    assert r.stdout.strip() == ""
    assert r.stderr.count("bench.py needs a GPU") >= 2, r.stderr[-2000:]
    assert "AssertionError" not in r.stderr
