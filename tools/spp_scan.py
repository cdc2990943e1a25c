#!/usr/bin/env python3
"""Time of one dispatch of the benchmark workload at 1, 2, 4 ... 64 samples per pixel per dispatch
(the viewer's call pattern is one sample per dispatch).
    python tools/spp_scan.py [spp,spp,... [key=value tuning settings ...]]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from pathtracer_amd import renderer as R
class A: subdivide = bench.DEFAULT_SUBDIVIDE; bvh = "host"
work = bench.build_workload(A)
g = R.Renderer(1920, 1080)
g.upload_nodes(work["nodes"]); g.upload_primitives(work["prims"]); g.upload_materials(work["mats"]); g.upload_environment(work["hdri"], work["cache"])
p = R.default_params(1920, 1080); p.max_bounces = 8; g.set_params(p)
if len(sys.argv) > 2:
    g.set_tuning(**{k: int(v) for k, v in (item.split("=") for item in sys.argv[2:])})
import time
for spp in ([int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else (1, 2, 4, 8, 16, 64)):
    g.clear(); g.dispatch(1, spp, 1); g.synchronize()
    n = max(4, 64 // spp)
    t0 = time.perf_counter(); rays = 0; f = 1
    for i in range(n):
        g.dispatch(f, spp, 1); f += spp
    g.synchronize()
    dt = time.perf_counter() - t0
    st = g.stats(); r = (st.rays_closest + st.rays_any)
    print(f"{spp:3d} spp per dispatch: {dt / n * 1e3:8.3f} ms per dispatch, {dt / n / spp * 1e3:7.3f} ms per spp, {r / (dt / n) / 1e6:8.1f} Mrays/s, device {g.last_dispatch_ms():.3f} ms")
