import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from pathtracer_amd import renderer as R
class A: subdivide = bench.DEFAULT_SUBDIVIDE; bvh = "host"
work = bench.build_workload(A)
g = R.Renderer(1920, 1080)
g.upload_nodes(work["nodes"]); g.upload_primitives(work["prims"]); g.upload_materials(work["mats"]); g.upload_environment(work["hdri"], work["cache"])
p = R.default_params(1920, 1080); p.max_bounces = 8; g.set_params(p)
g.clear(); g.dispatch(1, 16, 1); g.clear(); g.dispatch(1, 16, 1, counted=True); g.synchronize()
pr = g.profile()
print("phases", pr[9], "with something", pr[6], "hit-block phases", pr[10], "miss phases", pr[11], "both", pr[15], "iterations", pr[0], "hit lanes waiting/phase", pr[7]/pr[9], "miss lanes/phase", pr[8]/pr[9])
