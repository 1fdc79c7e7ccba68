"""The driver's contract for bench.py (task statement): ONE JSON line on stdout with the prescribed keys, launched
exactly as the driver launches it -- plain for N = 1, through `python -m torch.distributed.run` for N > 1 (here two
ranks sharing the box's single GPU over gloo, QS_BENCH_SHARE_GPU=1, which drives the whole N > 1 code path: rendezvous,
statistics exchange every step, barrier + max-over-ranks timing, rank-0-only output, config 5 under DDP)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
        "data", "config", "roofline"}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _check(rec, n, steps):
    assert KEYS <= set(rec), sorted(KEYS - set(rec))
    baseline = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert rec["metric"] == baseline["metric"] and rec["unit"] == "Gelem/s"
    assert rec["n_gpus"] == n and rec["steps"] == steps and rec["scaling"] == "weak" and rec["higher_is_better"] is True
    assert rec["vs_baseline"] is None and rec["dtype"] == "f32" and rec["data"] == "synthetic"
    numel = 256 * 256 * 56 * 56
    assert abs(rec["value"] - n * numel / (rec["ms_per_step"] * 1e-3) / 1e9) < 0.01 * rec["value"]      # whole-job aggregate
    roof = rec["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert 0 < roof["frac"] <= 1.0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    for k in roof["kernels"].values():
        assert 0 < k["frac"] <= 1.0        # mask-aware byte counts: no kernel "beats" the roofline
    assert "workload" in rec["config"] and "model" not in rec["config"]


def test_single_gpu_line_with_roofline_variants_and_cpu_baseline():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "24", "--warmup", "2", "--no-configs"],
                       capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    rec = json.loads(lines[0])
    _check(rec, 1, 24)
    assert rec["config"]["elide_pruned"] == "forward" and set(rec["config"]["variants"]) == {"off", "all", "frozen_mask"}
    # (orderings of TIMES are only checked where the byte counts differ by a wide margin -- 9.5 against 14 B/elem -- and between
    #  figures taken back to back; the 24 timed headline steps of this short run can fall into the clock ramp of a cold GPU)
    v = rec["config"]["variants"]
    assert 0 < v["all"]["ms_per_step"] < v["off"]["ms_per_step"] and 0 < rec["ms_per_step"] < 2.0 * v["off"]["ms_per_step"]
    cpu = rec["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and 0 < cpu["value"] < rec["value"]
    assert rec["roofline"]["algorithmic_bytes_per_launch"] in (6 * 256 * 256 * 56 * 56,
                                                               int(round(rec["config"]["algorithmic_bytes_per_elem"]["apply_fwd"] * 256 * 256 * 56 * 56)))


def test_two_ranks_through_torch_distributed_run():
    env = dict(os.environ, QS_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", QS_BENCH_DDP_BATCH="8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                # rank 0 only
    rec = json.loads(lines[0])
    _check(rec, 2, 12)
    assert "exchange" in rec["config"] and "cpu_baseline" not in rec
    assert rec["config"]["ranks_seen"] == 2 and rec["config"]["backend"] == "gloo"      # (an all-reduce of ones: the group saw both ranks)
    # BASELINE config 5 rides on the N > 1 line: ResNet-50 --pq under DistributedDataParallel (here batch 8 per rank)
    c5 = rec["configs"]["config5_resnet50_ddp"]
    assert "error" not in c5, c5
    assert c5["world"] == 2 and c5["input_shape_per_gpu"] == [8, 3, 224, 224]
    assert c5["pq_images_per_s"] > 0 and c5["plain_images_per_s"] > 0
    assert c5["operator_state_identical_across_ranks"] is True


def test_two_ranks_without_a_launcher():
    """`python bench.py --gpus 2` as the docstring advertises it: no torchrun, WORLD_SIZE unset -- bench.py starts its own ranks
    (before anything touches the GPU in the parent) and relays rank 0's single line and the exit code"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(QS_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", QS_BENCH_NO_DDP_CONFIG="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "2"],
                       capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    _check(rec, 2, 8)
    assert rec["config"]["ranks_seen"] == 2


def test_config5_watchdog_keeps_the_headline_marks_it_degraded_and_fails_the_run():
    """a rank that hangs in config 5's collectives must not cost the headline record, and must not look like success:
    the line carries a top-level `degraded: true` and every rank exits non-zero (here the deadline is simply too short)"""
    env = dict(os.environ, QS_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", QS_BENCH_DDP_BATCH="8",
               QS_BENCH_DDP_TIMEOUT="0.5")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert r.returncode != 0
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    _check(rec, 2, 8)
    assert rec["degraded"] is True and "error" in rec["configs"]["config5_resnet50_ddp"]
