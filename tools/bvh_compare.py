#!/usr/bin/env python3
"""Host sweep-SAH builder against the GPU linear-BVH builder on the benchmark scene and on
instanced copies of it: build time, node count, rays/s and visits per ray of the same render.

    python tools/bvh_compare.py [copies ...]      (default: 1 9 36)
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd import renderer as R  # noqa: E402
from pathtracer_amd import scenes  # noqa: E402


def instanced_soup(copies):
    meshes, mats, _ = scenes.instanced_bunnies(copies)
    return np.concatenate([R.assemble_primitives(v, i, k) for k, (v, i) in enumerate(meshes)]), mats


def render(g, mats, hdri, cache, width, height, spp, side):
    g.upload_materials(mats)
    g.upload_environment(hdri, cache)
    p = R.default_params(width, height)
    p.max_bounces = 8
    R.set_camera(p, (-2.5 - 1.1 * side, 1.0 + 0.8 * side, 0.0), (0.0, 0.0, 0.0), width, height)
    g.set_params(p)
    g.clear()
    g.dispatch(1, spp, 1)  # warm-up: also measures the tile costs the timed launch is planned from
    g.synchronize()
    g.clear()
    g.dispatch(1, spp, 1)
    g.synchronize()
    ms = g.last_dispatch_ms()
    g.clear()
    g.dispatch(1, spp, 1, counted=True)  # the counting build of the same launch, for the per-ray figures only
    g.synchronize()
    st = g.stats()
    rays = st.rays_closest + st.rays_any
    return rays / ms / 1e3, st.node_visits / rays, st.triangle_tests / rays, st.max_stack_depth


def main():
    copies = [int(x) for x in sys.argv[1:]] or [1, 9, 36]
    hdri = scenes.sky_hdri(1024, 512)
    cache = R.calculate_hdri_cache(hdri)
    width, height, spp = 1920, 1080, 16
    for c in copies:
        prims100, mats = instanced_soup(c)
        side = int(np.ceil(np.sqrt(c)))
        n = prims100.shape[0]
        print(f"{c:4d} bunnies, {n:9d} triangles", flush=True)
        for label, full in (("host sweep SAH (the reference's tree)", False), ("host sweep SAH without the cost cap", True)):
            if n > (25_000_000 if full else 6_000_000):
                continue
            t0 = time.time()
            sorted100, nodes40 = R.build_bvh_sah(prims100, 4, full_sweep=full)
            t_host = time.time() - t0
            g = R.Renderer(width, height)
            g.upload_nodes(R.align_nodes(nodes40))
            g.upload_primitives(R.align_primitives(sorted100))
            mr, nv, tt, sd = render(g, mats, hdri, cache, width, height, spp, side)
            print(f"     {label:38s} {t_host:7.2f} s, {mr:7.0f} Mrays/s, {nv:5.1f} visits + {tt:4.1f} tests per ray, stack {sd}", flush=True)
            g.close()
        aligned = R.align_primitives(prims100)
        for label, radius in (("GPU Morton radix tree", 0), ("GPU locally-ordered clustering", -1)):
            g = R.Renderer(width, height)
            if radius >= 0:
                g.set_tuning(ploc_radius=radius)
            g.build_bvh_device(aligned[:1000])  # warm-up: module load, hipcub first use
            t0 = time.time()
            g.build_bvh_device(aligned)
            t_dev = time.time() - t0
            mr, nv, tt, sd = render(g, mats, hdri, cache, width, height, spp, side)
            print(f"     {label:38s} {t_dev:7.3f} s (incl. {aligned.nbytes / 1e6:.0f} MB upload), {mr:7.0f} Mrays/s, {nv:5.1f} visits + "
                  f"{tt:4.1f} tests per ray, stack {sd}", flush=True)
            g.close()


if __name__ == "__main__":
    main()
