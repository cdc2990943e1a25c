#!/usr/bin/env python3
"""Steady-state rates of the other benchmark configurations (bench.py times only the headline one):
the reference's default scene at its default size, the headline scene at the reference's 3 bounces,
a textured multi-material scene, and the many-material sphere field at 16 bounces.  One warm-up
launch, then the median of three timed launches each; prints one line per configuration.

    python tools/config_table.py [--spp 64]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd import recipes, scenes  # noqa: E402
from pathtracer_amd import renderer as R  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--spp", type=int, default=64)
ap.add_argument("--bvh", choices=["host", "full-sweep", "device"], default="host",
                help="host: the reference's builder; full-sweep: the same without its cost cap; device: GPU linear BVH")
ap.add_argument("--no-rebuild", action="store_true", help="traverse the host builder's tree itself (pt_set_bvh_rebuild off)")
ap.add_argument("--depths", action="store_true", help="also print the share of paths alive at each depth (counting build)")
ap.add_argument("--tuning", default="", help="scheduler settings: key=value,key=value (Renderer.set_tuning)")
args = ap.parse_args()

hdri = scenes.sky_hdri(1024, 512)
cache = R.calculate_hdri_cache(hdri)


def run(name, width, height, bounces, meshes, mats, textures=(), eye=None, look_at=(0.0, 0.0, 0.0)):
    g = R.Renderer(width, height)
    if args.tuning:  # before the scene goes in: some settings steer the device BVH builder
        g.set_tuning(**{k: int(v) for k, v in (item.split("=") for item in args.tuning.split(","))})
    for slot, texels, srgb in textures:
        g.upload_texture(slot, texels, srgb)
    if args.bvh == "device" or args.no_rebuild:
        for (v, i), m in zip(meshes, mats):
            g.AddPathTracingModel(v, i, m)
        if args.bvh == "device":
            g.ProcessBuffersDevice()
        else:
            g.ProcessBuffers(full_sweep=args.bvh == "full-sweep")  # (a tree of a builder named by the caller is traversed as built)
    else:
        # what the reference's host does: its own builder's tree, uploaded (the library then builds its own: the default)
        soup = np.concatenate([R.assemble_primitives(v, i, k) for k, (v, i) in enumerate(meshes)])
        sorted_prims, nodes40 = R.build_bvh_sah(soup, 4, full_sweep=args.bvh == "full-sweep")
        g.upload_nodes(R.align_nodes(nodes40))
        g.upload_primitives(R.align_primitives(sorted_prims))
        g.upload_materials(np.array(list(mats), dtype=scenes.MATERIAL_DTYPE))
    g.upload_environment(hdri, cache)
    p = R.default_params(width, height)
    p.max_bounces = bounces
    if eye is not None:
        R.set_camera(p, eye, look_at, width, height)
    g.set_params(p)
    g.clear()
    g.dispatch(1, args.spp, 1)  # warm-up: also measures the tile costs the next launches are ordered by
    g.synchronize()
    ms, rays = [], 0
    for _ in range(3):
        g.clear()
        g.dispatch(1, args.spp, 1, counted=False)
        g.synchronize()
        ms.append(g.last_dispatch_ms())
        st = g.stats()
        rays = st.rays_closest + st.rays_any
    t = float(np.median(ms))
    tris = sum(i.size // 3 for _, i in meshes)
    # wave utilisation from the counting build of the same launch
    g.clear()
    g.dispatch(1, args.spp, 1, counted=True)
    g.synchronize()
    st, pr = g.stats(), g.profile()
    depth = g.depth_histogram()
    print(f"{name:34s} {tris:8d} triangles {width}x{height} {bounces:2d} bounces {args.spp} spp: {t:8.2f} ms  "
          f"{rays / t / 1e3:8.1f} Mrays/s  {args.spp / t * 1e3:8.1f} spp/s  {rays / (width * height * args.spp):.2f} rays/sample  "
          f"{st.node_visits / rays:5.1f} visits + {st.triangle_tests / rays:4.1f} tests per ray  "
          f"lanes per traversal iteration {pr[1] / max(pr[0], 1):4.1f}, per hit block {pr[7] / max(pr[6], 1):4.1f}")
    if args.depths:  # share of the samples whose path is still alive at each depth (closest-hit queries per sample)
        alive = [d / max(depth[0], 1) for d in depth[:bounces + 1]]
        print(" " * 34 + " paths alive by depth: " + " ".join(f"{a:.3f}" for a in alive if a > 0))


m, a = scenes.flatten(scenes.recipe("BunnyClearcoat"))
run("BunnyClearcoat (reference default)", 512, 512, 3, m, a)
run("BunnyClearcoat", 1920, 1080, 3, m, a)
m, a = scenes.flatten(scenes.recipe("DragonGlass"))
v, i = scenes.subdivide(*m[0], 1)
run("DragonGlass stand-in, 3 bounces", 1920, 1080, 3, [(v, i), m[1]], a)
run("DragonGlass stand-in, 8 bounces", 1920, 1080, 8, [(v, i), m[1]], a)
# the same scene from close by: the glass mesh fills most of the frame (the reference camera sees mostly floor and sky)
run("DragonGlass stand-in, close-up", 1920, 1080, 8, [(v, i), m[1]], a, eye=(-0.15, 0.8, 0.85), look_at=(-0.15, 0.8, 0.0))
m, a, tex, _ = recipes.load_scene("Mill")
run("Mill (3 textures)", 1920, 1080, 8, m, a, tex, eye=(14.0, 9.0, 16.0), look_at=(0.0, 5.0, 0.0))
m, a, tex, _ = recipes.load_scene("AreaLight")
run("AreaLight (emissive + transmissive)", 1920, 1080, 8, m, a, tex, eye=(0.5, 3.0, -8.0), look_at=(0.0, 1.2, 2.0))
m, a = scenes.sphere_field()
run("Sphere field, 65 materials", 1920, 1080, 16, m, a, eye=(-14.0, 7.0, 0.0))
