#!/usr/bin/env python3
"""One-GPU rehearsal of the multi-GPU data path: a 1-rank RCCL group, the kernel writing into the
torch-owned accumulation tensor on the shared stream, and the sum all-reduce issued on that tensor.
(With one rank the all-reduce is an identity; the point is that every call of the N > 1 path runs.)"""
import faulthandler
import os
import sys

faulthandler.dump_traceback_later(120, exit=True)  # a blocked rank ends with its stacks on stderr
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")  # tests pass a free port
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from pathtracer_amd import renderer as R  # noqa: E402
from pathtracer_amd.distributed import ShardedRenderer  # noqa: E402
from tests.common import built_scene, environment  # noqa: E402

torch.cuda.set_device(0)
import datetime  # noqa: E402
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0), timeout=datetime.timedelta(seconds=90))
nodes, prims, mats = built_scene("TorusMixed")
hdri, cache = environment(256, 128)
w, h, spp = 256, 192, 8
from pathtracer_amd import distributed as D  # noqa: E402
D.reduce_sum = lambda t, group=None: dist.all_reduce(t, op=dist.ReduceOp.SUM)  # even with one rank
sr = ShardedRenderer(w, h, 0, 0, 2)  # world 2: sum mode, separate total buffer, collective stream
g = sr.renderer
g.upload_nodes(nodes); g.upload_primitives(prims); g.upload_materials(mats); g.upload_environment(hdri, cache)
p = R.default_params(w, h)
sr.set_params(p)
sr.clear()
half = spp // 2
g.dispatch(1, half, 1)          # two steps with a reduce after each, like the benchmark loop
sr.reduce()
g.dispatch(1 + half, spp - half, 1)
sr.reduce()
out = sr.resolve(spp).cpu().numpy()
# reference: the plain single-GPU running mean of the same frames
ref = R.Renderer(w, h)
ref.upload_nodes(nodes); ref.upload_primitives(prims); ref.upload_materials(mats); ref.upload_environment(hdri, cache)
p2 = R.default_params(w, h)
ref.set_params(p2)
ref.clear()
ref.dispatch(1, spp, 1)
a = ref.read_image(0)
err = np.abs(out[0][..., :3] - a[..., :3]).max() / max(a[..., :3].max(), 1e-9)
print(f"RCCL 1-rank rehearsal: max relative difference sum/N vs running mean = {err:.2e}")
assert err < 1e-5
dist.destroy_process_group()
print("ok")
