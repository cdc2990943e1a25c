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


def test_pmc_summary_is_made_of_the_committed_counters_and_found_by_launch_shape(tmp_path):
    """tools/make_pmc_summary.py on the committed rocprofv3 counter summary of the benchmark launch, and
    bench.py's lookup of the result by launch shape."""
    import argparse
    out = tmp_path / "pmc_summary.json"
    shape = "scene=dragonglass width=1920 height=1080 spp_per_step=64 max_bounces=8 subdivide=1 bvh=host copies=1 rebuild=1".split()
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_pmc_summary.py"), os.path.join("profiles", "r03_c2_pmc.json"),
                    str(out)] + shape, check=True, cwd=ROOT, capture_output=True)
    t = json.loads(out.read_text())
    assert 0.5 < t["valu_busy"] <= 1.0 and 0.2 < t["lane_utilisation"] < 1.0 and 2.0 < t["waves_per_simd"] <= 3.0
    assert 0.5 < t["l1_busy"] < 1.0 and 0.9 < t["l1_busy"] + t["l1_stall"] < 1.1 and t["bound"] == "hbm"
    assert t["bytes_per_launch"] == (2.0 * t["fetch_size_kb"] + t["write_size_kb"]) * 1024.0 and "3 waves per SIMD" in t["limited_by"]
    assert t["valu_busy_raw"] >= t["valu_busy"]
    committed = json.load(open(os.path.join(ROOT, "profiles", "pmc_summary.json")))
    keys = ("scene", "width", "height", "spp_per_step", "max_bounces", "subdivide", "bvh", "copies", "rebuild")
    assert {k: committed[k] for k in keys} == {k: t[k] for k in keys}
    sys.path.insert(0, ROOT)
    import bench
    args = argparse.Namespace(scene="dragonglass", bounces=8, subdivide=1, bvh="host", copies=1, no_rebuild=False)
    assert bench.pmc_summary(args, 1920, 1080, 64)["bytes_per_launch"] == committed["bytes_per_launch"]
    assert bench.pmc_summary(args, 1280, 720, 64) is None  # another launch shape: the counters do not apply
    args.no_rebuild = True  # the uploaded tree traversed as it is: another kernel workload, the counters do not apply either
    assert bench.pmc_summary(args, 1920, 1080, 64) is None
