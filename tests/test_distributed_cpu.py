"""Sample-sharded multi-GPU path, rehearsed on CPU: two gloo ranks render disjoint FrameCount
shards (with the CPU oracle standing in for the GPU renderer), sum-reduce, and must reproduce the
single-process sample set."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pathtracer_amd.distributed import reduce_sum, shard_frames


def test_shards_partition_the_sample_set():
    for world in (1, 2, 3, 4, 8):
        for spp in (1, 2, 7, 8, 64, 1024):
            frames = []
            for r in range(world):
                first, count, stride = shard_frames(r, world, spp)
                frames += [first + k * stride for k in range(count)]
            assert sorted(frames) == list(range(1, spp + 1)), (world, spp)
    assert shard_frames(3, 4, 2) == (4, 0, 4)  # more ranks than samples: nothing to do
    assert shard_frames(1, 2, 5, first_frame=11) == (12, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, spp, width, height, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # both ranks are on this host; do not resolve the host name
    import datetime
    import faulthandler
    faulthandler.dump_traceback_later(150, exit=True)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    from oracle import binding as ob
    from pathtracer_amd import renderer as R
    from tests.common import built_scene, environment
    nodes, prims, mats = built_scene("TorusMixed")
    hdri, cache = environment(64, 32)
    o = ob.Oracle()
    o.set_scene(nodes, prims, mats)
    o.set_environment(hdri, cache)
    p = R.default_params(width, height)
    p.accumulate_mode = 1  # sums, like ShardedRenderer on N > 1 GPUs
    first, count, stride = shard_frames(rank, world, spp)
    imgs = [np.zeros((height, width, 4), np.float32) for _ in range(3)]
    if count:
        o.render(p, width, height, first, count, stride, images=imgs, threads=2)
    stacked = torch.from_numpy(np.stack(imgs))
    reduce_sum(stacked)
    if rank == 0:
        np.save(os.path.join(out_dir, "total.npy"), stacked.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("spp", [5, 8])
def test_two_rank_gloo_reduce_equals_single_process(tmp_path, spp):
    width, height, world = 40, 32, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, spp, width, height, str(tmp_path)), nprocs=world, join=True)
    total = np.load(tmp_path / "total.npy")

    from oracle import binding as ob
    from pathtracer_amd import renderer as R
    from tests.common import built_scene, environment
    nodes, prims, mats = built_scene("TorusMixed")
    hdri, cache = environment(64, 32)
    o = ob.Oracle()
    o.set_scene(nodes, prims, mats)
    o.set_environment(hdri, cache)
    p = R.default_params(width, height)
    p.accumulate_mode = 1
    single = o.render(p, width, height, 1, spp, 1)
    # same samples, different float summation order
    for k in range(3):
        assert np.all(total[k][..., 3] == float(spp))  # every pixel got every sample exactly once
        assert np.allclose(total[k], single[k], rtol=1e-5, atol=1e-6)
    # and the mean agrees with the reference's running mean to rounding
    p.accumulate_mode = 0
    mean = o.render(p, width, height, 1, spp, 1)[0]
    assert np.allclose(total[0][..., :3] / spp, mean[..., :3], rtol=2e-5, atol=1e-6)
