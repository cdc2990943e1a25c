"""The multi-GPU data path on hardware.  Child processes only: a process that has touched the GPU
must not be replaced by another program, and a process group must not outlive its test."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["MASTER_ADDR"] = "127.0.0.1"
    env["MASTER_PORT"] = str(_free_port())
    return env


def test_sharded_renderer_streams_and_collective_on_one_gpu():
    """ShardedRenderer's N > 1 path (sum mode, separate total buffer, snapshot on the render stream, RCCL
    all-reduce on a second stream chained by events) driven on one GPU with a 1-rank RCCL group."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_rccl_single.py")], capture_output=True, text=True,
                       timeout=300, env=_env(), cwd=ROOT)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-1500:] + r.stderr[-1500:]


def test_two_ranks_reduce_equals_single_gpu_render():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    env = _env()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", env["MASTER_PORT"], os.path.join(ROOT, "tools", "check_rccl_ranks.py")],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


def test_bench_two_gpus_self_launch():
    import json
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--spp-per-step", "8", "--width", "640", "--height", "360", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env=_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["value"] > 0


def test_bench_two_ranks_share_one_gpu_over_gloo():
    """bench.py's N > 1 control flow (self-launch, sharding by FrameCount, sum mode, the reduce inside the timed
    region, max-over-ranks timing) on however many GPUs there are: with PT_BENCH_BACKEND=gloo the ranks share
    the devices (RCCL takes one rank per GPU).  A rehearsal, not a measurement."""
    import json
    env = _env()
    env["PT_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--spp-per-step", "8", "--width", "640", "--height", "360", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["config"]["total_spp"] == 2 * 8 * 2
    assert "REHEARSAL" in out["config"]["reduce"]
