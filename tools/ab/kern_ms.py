import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from pathtracer_amd import renderer as R
class A: subdivide = bench.DEFAULT_SUBDIVIDE; bvh = "host"
work = bench.build_workload(A)
g = R.Renderer(1920, 1080)
for item in sys.argv[1:]:
    k, v = item.split("="); g.set_tuning(**{k: int(v)})
g.upload_nodes(work["nodes"]); g.upload_primitives(work["prims"]); g.upload_materials(work["mats"]); g.upload_environment(work["hdri"], work["cache"])
p = R.default_params(1920, 1080); p.max_bounces = 8; g.set_params(p)
g.clear(); g.dispatch(1, 4, 1); g.synchronize()
res = []
for r in range(4):
    g.clear(); g.dispatch(1, 64, 1); g.synchronize(); res.append((g.last_dispatch_ms(),) + tuple(g.last_kernel_ms()))
print(os.environ.get("PT_LIBPTCORE", "default"), sys.argv[1:], "dispatch/trace/resolve ms:", [tuple(round(x, 3) for x in t) for t in res])
