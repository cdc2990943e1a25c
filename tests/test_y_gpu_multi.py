"""The multi-GPU data path on hardware.  Child processes only: a process that has touched the GPU
must not be replaced by another program.  Every child runs in a session of its own under a time
limit far below the suite's; on expiry the whole process group is killed and both output streams go
into the assertion message, so a blocked rank fails one test instead of stalling the run.  The file
sorts after the parity tests on purpose: these are rehearsals of control flow."""
import json
import os
import signal
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH_ARGS = ["--steps", "2", "--warmup", "1", "--spp-per-step", "8", "--width", "640", "--height", "360", "--no-cpu-baseline"]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["MASTER_ADDR"] = "127.0.0.1"
    env["MASTER_PORT"] = str(_free_port())
    env["GLOO_SOCKET_IFNAME"] = "lo"  # every rank is on this host; do not resolve the host name
    env.update(extra)
    return env


def run_group(cmd, env, limit):
    """Run `cmd` as the leader of a new session; kill the whole group when `limit` seconds pass or when the
    leader ends with ranks still alive.  -> (return code or None on expiry, stdout, stderr)."""
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT,
                         start_new_session=True)
    try:
        out, err = p.communicate(timeout=limit)
        rc = p.returncode
    except subprocess.TimeoutExpired:
        rc = None
    finally:
        try:
            os.killpg(p.pid, signal.SIGKILL)  # the session leader's pid is the group's id
        except ProcessLookupError:
            pass
    if rc is None:
        try:
            out, err = p.communicate(timeout=10)
        except subprocess.TimeoutExpired:
            out, err = "", "(output pipes still held after the group was killed)"
    return rc, out, err


def _report(rc, out, err, limit):
    head = f"timed out after {limit} s (process group killed)" if rc is None else f"exit code {rc}"
    return f"{head}\n--- stdout\n{out[-3000:]}\n--- stderr\n{err[-6000:]}"


def _rendezvous_blocked(err):
    """True when a rank's watchdog dump (or gloo itself) shows it never got past connecting to its peer: the
    host's loopback rendezvous is not usable, which is not a property of the code under rehearsal."""
    marks = (" in init_process_group", " in _new_process_group_helper", " in _store_based_barrier",  # faulthandler frames
             "connectFullMesh", "Timed out waiting for clients", "DistNetworkError", "DistStoreError")
    return any(m in err for m in marks)


def test_sharded_renderer_streams_and_collective_on_one_gpu():
    """ShardedRenderer's N > 1 path (sum mode, separate total buffer, snapshot on the render stream, RCCL
    all-reduce on a second stream chained by events) driven on one GPU with a 1-rank RCCL group."""
    rc, out, err = run_group([sys.executable, os.path.join(ROOT, "tools", "check_rccl_single.py")], _env(), 150)
    assert rc == 0 and out.strip().endswith("ok"), _report(rc, out, err, 150)


def test_two_ranks_reduce_equals_single_gpu_render():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    env = _env()
    rc, out, err = run_group([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                              "127.0.0.1", "--master-port", env["MASTER_PORT"], os.path.join(ROOT, "tools", "check_rccl_ranks.py")],
                             env, 300)
    assert rc == 0 and "ok" in out, _report(rc, out, err, 300)


def test_bench_two_gpus_self_launch():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    rc, out, err = run_group([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + BENCH_ARGS,
                             _env(PT_BENCH_WATCHDOG_S="240"), 300)
    assert rc == 0, _report(rc, out, err, 300)
    line = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["rank_ms_min"] <= line["rank_ms_max"]


def test_bench_two_ranks_share_one_gpu_over_gloo():
    """bench.py's N > 1 control flow (self-launch, sharding by FrameCount, sum mode, the reduce inside the timed
    region, max-over-ranks timing) on however many GPUs there are: with PT_BENCH_BACKEND=gloo the ranks share
    the devices and the collectives go through gloo on host tensors (RCCL takes one rank per GPU).  A
    rehearsal, not a measurement: it covers everything but the RCCL calls themselves.  Each rank dumps its
    stacks and exits after 90 s (bench.py's watchdog); the group is killed at 120 s."""
    rc, out, err = run_group([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + BENCH_ARGS,
                             _env(PT_BENCH_BACKEND="gloo", PT_BENCH_WATCHDOG_S="90", TORCH_DISTRIBUTED_DEBUG="DETAIL"), 120)
    if rc != 0 and _rendezvous_blocked(err):
        pytest.skip("gloo rendezvous over the loopback interface did not complete on this host:\n" + _report(rc, out, err, 120))
    assert rc == 0, _report(rc, out, err, 120)
    line = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["config"]["total_spp"] == 2 * 8 * 2
    assert "REHEARSAL" in line["config"]["reduce"]
    assert line["rank_ms_min"] <= line["rank_ms_max"] and line["ms_per_step_median"] > 0
