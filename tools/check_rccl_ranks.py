#!/usr/bin/env python3
"""Multi-GPU data path on real ranks: started under torch.distributed.run with one process per GPU,
every rank renders its FrameCount shard through ShardedRenderer (sum mode, kernel writing into the
torch-owned accumulation tensor, snapshot + RCCL all-reduce on a second stream), and rank 0 checks
reduce() + resolve() against a plain single-GPU running mean of all the frames.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 tools/check_rccl_ranks.py"""
import faulthandler
import os
import sys

faulthandler.dump_traceback_later(240, exit=True)  # a blocked rank ends with its stacks on stderr
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from pathtracer_amd import renderer as R  # noqa: E402
from pathtracer_amd.distributed import ShardedRenderer  # noqa: E402
from tests.common import built_scene, environment  # noqa: E402

rank, local, world = int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(local)
import datetime  # noqa: E402
dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=datetime.timedelta(seconds=180))
nodes, prims, mats = built_scene("TorusMixed")
hdri, cache = environment(256, 128)
w, h, spp_per_rank, steps = 256, 192, 3, 2
sr = ShardedRenderer(w, h, local, rank, world)
g = sr.renderer
g.upload_nodes(nodes); g.upload_primitives(prims); g.upload_materials(mats); g.upload_environment(hdri, cache)
p = R.default_params(w, h)
sr.set_params(p)
sr.clear()
for step in range(steps):  # a reduce after every step: the second one waits for the first collective's event
    sr.render_step(step, spp_per_rank)
    sr.reduce()
total_spp = steps * spp_per_rank * world
out = sr.resolve(total_spp).cpu().numpy()
if rank == 0:
    ref = R.Renderer(w, h, device=local)
    ref.upload_nodes(nodes); ref.upload_primitives(prims); ref.upload_materials(mats); ref.upload_environment(hdri, cache)
    ref.set_params(R.default_params(w, h))
    ref.clear()
    ref.dispatch(1, total_spp, 1)
    worst = 0.0
    for k in range(3):
        a = ref.read_image(k)
        worst = max(worst, float(np.abs(out[k][..., :3] - a[..., :3]).max() / max(a[..., :3].max(), 1e-9)))
    print(f"RCCL {world}-rank check: max relative difference sum/N vs running mean = {worst:.2e}")
    assert worst < 1e-5  # same samples, float32 sums in another order
dist.barrier()
dist.destroy_process_group()
if rank == 0:
    print("ok")
