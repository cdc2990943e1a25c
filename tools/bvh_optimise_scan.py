#!/usr/bin/env python3
"""Reinsertion passes of the device builder (tuning keys bvh_optimise, bvh_optimise_mod) on a benchmark scene:
build time, validity and surface-area cost of the tree built, and the speed of a launch traced through it from
the scene's camera and from the close-up.
    python tools/bvh_optimise_scan.py [--scene dragonglass] [--spp 16] [--check] 0:8 8:8 16:8 ...   (passes:mod)
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench  # noqa: E402
from pathtracer_amd import renderer as R  # noqa: E402
from tests.common import check_tree, sah_cost  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("settings", nargs="*", default=["0:8", "8:8"])
ap.add_argument("--scene", default="dragonglass")
ap.add_argument("--copies", type=int, default=1)
ap.add_argument("--spp", type=int, default=16)
ap.add_argument("--check", action="store_true", help="walk the downloaded tree on the host (slow beyond a million triangles)")
ap.add_argument("--cost", action="store_true", help="surface-area cost of the downloaded tree")
ap.add_argument("--tuning", default="", help="other tuning settings for every build, key=value,key=value")
ap.add_argument("--width", type=int, default=0)
ap.add_argument("--height", type=int, default=0)
a = ap.parse_args()


class A:
    scene = a.scene
    copies = a.copies
    subdivide = bench.DEFAULT_SUBDIVIDE if a.scene in ("dragonglass", "closeup") and a.copies == 1 else 0
    bvh = "device"


_, W, H, bounces = bench.SCENES[a.scene]
W, H = a.width or W, a.height or H
work = bench.build_workload(A)
source = work["prims"]
views = [("scene camera", bench.scene_params(work, W, H, bounces))]
if a.scene == "dragonglass" and a.copies == 1:
    p2 = R.default_params(W, H)
    p2.max_bounces = bounces
    R.set_camera(p2, (-0.15, 0.8, 0.85), (-0.15, 0.8, 0.0), W, H)
    views.append(("close-up", p2))
print(f"{a.scene}: {source.shape[0]} triangles, {W}x{H}, {a.spp} spp per launch", flush=True)
g = R.Renderer(W, H)
for slot, texels, srgb in work.get("textures", []):
    g.upload_texture(slot, texels, srgb)
if a.tuning:
    g.set_tuning(**{k: int(v) for k, v in (item.split("=") for item in a.tuning.split(","))})
g.build_bvh_device(source[:1000])  # warm-up: module load, hipcub's first use
for setting in a.settings:
    passes, mod = (int(x) for x in setting.split(":"))
    g.set_tuning(bvh_optimise=passes, bvh_optimise_mod=mod)
    t0 = time.time()
    g.build_bvh_device(source)
    g.synchronize()
    seconds = time.time() - t0
    line = f"passes {passes:3d} mod {mod:2d}: build {seconds * 1e3:7.1f} ms"
    if a.check or a.cost:
        nodes, prims = g.download_bvh()
        if a.check:
            depth, interior = check_tree(nodes, prims, source)
            line += f", valid (depth {depth}, {interior} interior nodes)"
        if a.cost:
            line += f", cost {sah_cost(nodes):8.2f}"
    g.upload_materials(work["mats"])
    g.upload_environment(work["hdri"], work["cache"])
    for name, params in views:
        g.set_params(params)
        g.clear()
        g.dispatch(1, a.spp, 1)
        g.synchronize()
        ms = []
        for _ in range(3):
            g.clear()
            g.dispatch(1, a.spp, 1)
            g.synchronize()
            ms.append(g.last_dispatch_ms())
        g.clear()
        g.dispatch(1, a.spp, 1, counted=True)
        g.synchronize()
        st = g.stats()
        rays = st.rays_closest + st.rays_any
        line += f" | {name}: {np.median(ms):7.2f} ms, {rays / np.median(ms) / 1e3:7.0f} Mrays/s, {st.node_visits / rays:5.2f} visits + {st.triangle_tests / rays:4.2f} tests"
    print(line, flush=True)
