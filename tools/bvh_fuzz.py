#!/usr/bin/env python3
"""Random triangle soups through the device BVH builder with random builder settings (clustering radius, reinsertion
passes and candidate spacing, leaf size); every tree that comes back is walked on the host: each triangle in exactly one
leaf, boxes tight, links a tree (tests/common.py check_tree).
    python tools/bvh_fuzz.py [seconds] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from pathtracer_amd import renderer as R  # noqa: E402
from tests.common import check_tree  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def soup(count, kind):
    """`count` triangles as (vertices (3 count, 8), indices)."""
    if kind == 0:    # small triangles on a sphere
        c = rng.normal(size=(count, 3)); c /= np.linalg.norm(c, axis=1, keepdims=True)
        size = 0.05
    elif kind == 1:  # a few clumps far apart, sizes over six orders of magnitude
        c = rng.normal(size=(count, 3)) * 0.01 + rng.integers(0, 4, size=(count, 1)) * 1000.0
        size = 10.0 ** rng.uniform(-4, 2, size=(count, 1, 1))
    elif kind == 2:  # many coincident triangles and a few others
        c = np.zeros((count, 3)); c[::7] = rng.normal(size=(len(c[::7]), 3))
        size = 1.0
    elif kind == 3:  # long thin triangles along one axis
        c = rng.uniform(-1, 1, size=(count, 3)) * np.array([100.0, 0.01, 0.01])
        size = np.array([50.0, 0.001, 0.001])
    else:            # a plane of small triangles under one huge one
        c = rng.uniform(-1, 1, size=(count, 3)) * np.array([1.0, 0.0, 1.0])
        size = 0.02
    tri = c[:, None, :] + rng.normal(size=(count, 3, 3)) * size
    if kind == 4:
        tri[0] = np.array([[-50, -0.1, -50], [50, -0.1, -50], [0, -0.1, 80]])
    v = np.zeros((3 * count, 8), np.float32)
    v[:, 0:3] = tri.reshape(-1, 3)
    v[:, 5] = 1.0
    return v, np.arange(3 * count, dtype=np.uint32)


t_end = time.time() + seconds
builds = 0
g = R.Renderer(32, 32)
while time.time() < t_end:
    count = int(rng.choice([1, 2, 3, 7, 8, 9, 33, 200, 1500, 6000, 20000]))
    kind = int(rng.integers(0, 5))
    v, idx = soup(count, kind)
    source = R.align_primitives(R.assemble_primitives(v, idx, 0))
    tuning = dict(ploc_radius=int(rng.choice([1, 4, 32, 200])), bvh_optimise=int(rng.choice([-1, 0, 1, 5, 40])),
                  bvh_optimise_mod=int(rng.choice([1, 2, 3, 8])), bvh_leaf_max=int(rng.choice([0, 1, 2, 3, 4])))
    g.set_tuning(**tuning)
    try:
        g.build_bvh_device(source)
        nodes, prims = g.download_bvh()
        check_tree(nodes, prims, source)
    except Exception as e:  # noqa: BLE001
        if "levels deep" in str(e):  # (a refusal the builder documents: the tree would not fit the traversal stacks)
            continue
        print(f"FAILED on {count} triangles of kind {kind} with {tuning}: {type(e).__name__}: {e}", flush=True)
        sys.exit(1)
    builds += 1
    if builds % 50 == 0:
        print(f"{builds} builds, all trees valid", flush=True)
print(f"fuzz done: {builds} builds, all trees valid")
