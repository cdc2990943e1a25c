#!/usr/bin/env python3
"""Wave-occupancy profile of the megakernel on the benchmark workload (counting build).
    python tools/occupancy_profile.py [spp] [key=value tuning settings ...]
PT_PROFILE_SCENE picks another of bench.py's scenes, PT_PROFILE_BVH the builder, PT_PROFILE_REBUILD=0 the uploaded tree itself."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import bench  # noqa: E402
from pathtracer_amd import renderer as R  # noqa: E402


class A:
    scene = os.environ.get("PT_PROFILE_SCENE", "dragonglass")  # any scene of bench.py
    copies = 1
    subdivide = bench.DEFAULT_SUBDIVIDE if scene in ("dragonglass", "closeup") else 0
    bvh = os.environ.get("PT_PROFILE_BVH", "host")  # host | full-sweep | device


spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
_, W, H, bounces = bench.SCENES[A.scene]
work = bench.build_workload(A)
g = R.Renderer(W, H)
g.set_bvh_rebuild(os.environ.get("PT_PROFILE_REBUILD", "1") == "1")
tuning = dict(item.split("=") for item in sys.argv[2:])
if tuning:  # (before the scene goes in: some settings steer the device builder)
    g.set_tuning(**{k: int(v) for k, v in tuning.items()})
bench.install_scene(g, work)
g.set_params(bench.scene_params(work, W, H, bounces))
g.clear()
g.dispatch(1, spp, 1)  # (leaves the tile costs the next launch is planned from)
g.clear()
g.dispatch(1, spp, 1, counted=True)
g.synchronize()
st, pr = g.stats(), g.profile()
rays = st.rays_closest + st.rays_any
print(f"resident waves of the kernel variant: {g.resident_waves()}")
print(f"rays {rays}  node visits/ray {st.node_visits / rays:.2f}  tri tests/ray {st.triangle_tests / rays:.2f}  ms {g.last_dispatch_ms():.2f}")
print(f"traversal iterations {pr[0]}  avg lanes traversing {pr[1] / pr[0]:.1f}")
print(f"  with node step   {pr[2]} ({pr[2] / pr[0]:.2%})  avg lanes {pr[3] / max(pr[2], 1):.1f}")
print(f"  with triangle step {pr[4]} ({pr[4] / pr[0]:.2%})  avg lanes {pr[5] / max(pr[4], 1):.1f}")
print(f"shading phases {pr[9]}  shading something {pr[6]}  hit lanes/phase {pr[7] / max(pr[6], 1):.1f}  miss lanes/phase {pr[8] / max(pr[6], 1):.1f}")
print(f"  finish lanes/phase {pr[10] / pr[9]:.1f}  new-sample lanes/phase {pr[11] / pr[9]:.1f}")
print(f"  traversal iterations per shading phase {pr[0] / pr[9]:.1f}")
print(f"  wave clock ticks: shading phases {pr[12]} ({pr[12] / max(pr[12] + pr[13], 1):.1%}), traversal phases {pr[13]}; "
      f"per shading phase {pr[12] / pr[9]:.0f}, per traversal iteration {pr[13] / pr[0]:.0f}")
print(f"  of the shading ticks: miss block {pr[15] / pr[12]:.1%}, hit block {pr[14] / pr[12]:.1%}, finish + new {1 - (pr[14] + pr[15]) / pr[12]:.1%}")
wt, khz = g.wave_times()
wt = wt[wt[:, 0] > 0]  # blocks past the end of the block list never start
if len(wt):
    t0 = wt[:, 0].min()
    start, end = (wt[:, 0] - t0) / khz, (wt[:, 1] - t0) / khz  # ms
    span = end.max()
    dur = end - start
    print(f"wave lifetimes: {len(wt)} waves, kernel span {span:.2f} ms, mean {dur.mean():.3f} ms, median {np.median(dur):.3f}, "
          f"p90 {np.percentile(dur, 90):.3f}, max {dur.max():.3f}; sum of lifetimes / span = {dur.sum() / span:.0f} waves resident on average")
    edges = np.linspace(0, span, 11)
    conc = [((start < b) & (end > a)).sum() for a, b in zip(edges[:-1], edges[1:])]
    print("  waves alive per tenth of the span:", conc)
    print("  wave ends, percentiles 1/10/25/50/75/90/99/100 in ms:", " ".join(f"{np.percentile(end, q):.3f}" for q in (1, 10, 25, 50, 75, 90, 99, 100)))
    print(f"  last wave to start began at {start.max():.2f} ms; waves still running at 90% of the span: {(end > 0.9 * span).sum()}")
