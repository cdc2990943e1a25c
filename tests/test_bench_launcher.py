"""bench.py's own launcher: `--gpus N` outside a torchrun environment must start N ranks and print
one line with n_gpus == N.  Run here on CPU with the stub renderer (gloo, no GPU)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub", "--width", "64", "--height", "32",
                        "--spp-per-step", "4", "--steps", "3", "--warmup", "1", *extra],
                       capture_output=True, text=True, timeout=300, env=env)
    return r


def test_gpus_2_self_launches_two_ranks():
    r = _run("--gpus", "2")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 alone prints
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["config"]["parallelism"] == "sample-sharded x2"
    # whole-job aggregate: both ranks' rays over the max-over-ranks time
    rays_per_rank = 2 * 64 * 32 * 4 * 3
    assert abs(out["value"] * 1e6 * out["ms_per_step"] * 3 / 1e3 - 2 * rays_per_rank) < 1.0
    assert out["config"]["total_spp"] == 2 * 3 * 4


def test_gpus_1_runs_in_process():
    r = _run()
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["roofline"]["frac"] > 0 and "frac_l2" in out["roofline"]


def test_more_gpus_than_devices_fails():
    env = {k: v for k, v in os.environ.items() if k != "WORLD_SIZE"}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4096"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode != 0 and "GPU(s) visible" in r.stderr
