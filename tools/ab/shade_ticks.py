"""Experiment (library built with -DPT_EXPERIMENT_SHADE_TIMERS): share of shadeRay's wave ticks by section.
    PT_LIBPTCORE=tools/ab/timers.so python tools/ab/shade_ticks.py [scene]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from pathtracer_amd import renderer as R, lib as L
class A:
    scene = sys.argv[1] if len(sys.argv) > 1 else "dragonglass"; copies = 1
    subdivide = bench.DEFAULT_SUBDIVIDE if scene in ("dragonglass", "closeup") else 0; bvh = "host"
_, W, H, bounces = bench.SCENES[A.scene]
work = bench.build_workload(A)
g = R.Renderer(W, H)
bench.install_scene(g, work)
g.set_params(bench.scene_params(work, W, H, bounces))
g.clear(); g.dispatch(1, 16, 1); g.synchronize()
lib = ctypes.CDLL(L.LIB_PATH)
out = (ctypes.c_ulonglong * 8)()
assert lib.pt_experiment_shade_ticks(out) == 0
g.clear(); g.dispatch(1, 64, 1); g.synchronize()
assert lib.pt_experiment_shade_ticks(out) == 0
t = list(out)
names = ["reads + hit attributes + light sample", "environment lookup", "material (textures)", "surface", "light evaluation (NEE)", "BSDF sample", "BSDF evaluation + roulette"]
tot = sum(t[:7])
print(A.scene, "ms", g.last_dispatch_ms())
for n, v in zip(names, t):
    print(f"  {n:42s} {v / tot:6.1%}  {v:.3e}")
if os.environ.get("PT_SHADE_TICKS_RAW"):
    print("  raw", t)
