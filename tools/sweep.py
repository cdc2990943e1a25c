#!/usr/bin/env python3
"""In-process A/B of scheduler settings on the benchmark workload (interleaved rounds).
    python tools/sweep.py --spp 16 --rounds 3 pipeline=0 pipeline=1 pipeline=0,refill_min=32
Each positional argument is one variant: comma-separated key=value tuning settings.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench  # noqa: E402
from pathtracer_amd import renderer as R  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("variants", nargs="*", default=["pipeline=0", "pipeline=1"])
ap.add_argument("--spp", type=int, default=16)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--bounces", type=int, default=8)
ap.add_argument("--subdivide", type=int, default=bench.DEFAULT_SUBDIVIDE)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--bvh", choices=["host", "full-sweep", "device"], default="host")
a = ap.parse_args()


class A:
    subdivide = a.subdivide
    bvh = a.bvh


work = bench.build_workload(A)
g = R.Renderer(a.width, a.height)
if work["nodes"] is None:
    g.build_bvh_device(work["prims"])
else:
    g.upload_nodes(work["nodes"]); g.upload_primitives(work["prims"])
g.upload_materials(work["mats"])
g.upload_environment(work["hdri"], work["cache"])
p = R.default_params(a.width, a.height)
p.max_bounces = a.bounces
g.set_params(p)
defaults = dict(pipeline=0, trav_exit=28, refill_min=16, check_every=2, traverse_blocks=0, hit_min=28, tile_order=1, tile_split=-1, sample_budget_mib=8192, pull_items=1)
variants = []
for v in a.variants:
    kw = dict(defaults)
    for item in v.split(","):
        k, val = item.split("=")
        kw[k] = int(val)
    variants.append((v, kw))
for _, kw in variants:  # warm every kernel
    g.set_tuning(**kw)
    g.dispatch(1, 1, 1)
g.synchronize()
res = {v: [] for v, _ in variants}
rays = {}
for r in range(a.rounds):
    for v, kw in variants:
        g.set_tuning(**kw)
        g.clear()
        g.dispatch(1, a.spp, 1)
        g.synchronize()
        res[v].append(g.last_dispatch_ms())
        st = g.stats()
        rays[v] = st.rays_closest + st.rays_any
for v, _ in variants:
    ms = np.array(res[v])
    print(f"{v:44s} median {np.median(ms):8.3f} ms  min {ms.min():8.3f} ms  -> {rays[v] / np.median(ms) / 1e3:8.1f} Mrays/s")
