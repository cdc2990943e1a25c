#!/usr/bin/env python3
"""In-process A/B sweep of the traversal-phase exit threshold on the benchmark workload.
    python tools/sweep.py [--spp 16] [--rounds 3] 8 16 24 32 40 48 56 64
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench  # noqa: E402
from pathtracer_amd import renderer as R  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("values", nargs="*", type=int, default=[16, 24, 32, 40, 48, 56])
ap.add_argument("--spp", type=int, default=16)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--bounces", type=int, default=8)
ap.add_argument("--subdivide", type=int, default=0)
a = ap.parse_args()


class A:
    subdivide = a.subdivide


work = bench.build_workload(A)
g = R.Renderer(1920, 1080)
g.upload_nodes(work["nodes"]); g.upload_primitives(work["prims"]); g.upload_materials(work["mats"])
g.upload_environment(work["hdri"], work["cache"])
p = R.default_params(1920, 1080)
p.max_bounces = a.bounces
g.set_params(p)
g.dispatch(1, 2, 1); g.synchronize()
res = {v: [] for v in a.values}
for r in range(a.rounds):
    for v in a.values:
        g.set_tuning(v)
        g.clear()
        g.dispatch(1, a.spp, 1)
        g.synchronize()
        res[v].append(g.last_dispatch_ms())
st = g.stats()
rays = st.rays_closest + st.rays_any
for v in a.values:
    ms = np.array(res[v])
    print(f"trav_exit={v:3d}/64  median {np.median(ms):8.3f} ms  min {ms.min():8.3f} ms  -> {rays / np.median(ms) / 1e3:8.1f} Mrays/s")
