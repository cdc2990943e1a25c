#!/usr/bin/env python3
"""Times alternative builds of libptcore.so on the benchmark workload, one subprocess per build
(a process can only load one copy of the library).  python tools/ab_libs.py [--bvh device] lib1.so lib2.so ..."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, os
sys.path.insert(0, %r)
import numpy as np
from pathtracer_amd import lib as L
L.LIB_PATH = sys.argv[1]
import bench
from pathtracer_amd import renderer as R
class A: subdivide = bench.DEFAULT_SUBDIVIDE; bvh = os.environ.get("PT_AB_BVH", "host")
work = bench.build_workload(A)
g = R.Renderer(1920, 1080)
g.set_bvh_rebuild(os.environ.get("PT_AB_REBUILD", "0") == "1")  # the uploaded tree itself unless asked otherwise
if work["nodes"] is None: g.build_bvh_device(work["prims"])
else: g.upload_nodes(work["nodes"]); g.upload_primitives(work["prims"])
g.upload_materials(work["mats"]); g.upload_environment(work["hdri"], work["cache"])
p = R.default_params(1920, 1080); p.max_bounces = 8; g.set_params(p)
g.dispatch(1, 2, 1); g.synchronize()
ms = []
for r in range(4):
    g.clear(); g.dispatch(1, 64, 1); g.synchronize(); ms.append(g.last_dispatch_ms())
st = g.stats(); rays = st.rays_closest + st.rays_any; waves = g.resident_waves()
g.clear(); g.dispatch(1, 4, 1, counted=True); sc = g.stats(); r4 = sc.rays_closest + sc.rays_any
print("%%-50s median %%8.3f ms  min %%8.3f  -> %%8.1f Mrays/s   %%.2f node visits + %%.2f triangle tests per ray, stack %%d, %%d waves resident, %%d wide levels" %% (os.path.basename(sys.argv[1]), np.median(ms), min(ms), rays / np.median(ms) / 1e3, sc.node_visits / r4, sc.triangle_tests / r4, sc.max_stack_depth, waves, g.tree_info()[0]))
''' % ROOT
libs = sys.argv[1:]
if libs[:1] == ["--bvh"]:
    os.environ["PT_AB_BVH"] = libs[1]
    libs = libs[2:]
for lib in libs:
    subprocess.run([sys.executable, "-c", CODE, os.path.abspath(lib)], check=True)
