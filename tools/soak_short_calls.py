#!/usr/bin/env python3
"""Soak of the overlapping short calls: random runs of dispatch calls of 1-9 frames with reads, clears, parameter
changes and tuning changes in between, each run repeated with the overlap off; the images must agree bit for bit.
    python tools/soak_short_calls.py [seconds] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from pathtracer_amd import renderer as R


class A:
    scene = "c1"; copies = 1; subdivide = 0; bvh = "host"


seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
W, H = 640, 360
work = bench.build_workload(A)
g = R.Renderer(W, H)
bench.install_scene(g, work)
params = [bench.scene_params(work, W, H, b) for b in (3, 6)]
t_end = time.time() + seconds
runs = calls = 0
while time.time() < t_end:
    plan = []
    f = 1
    for _ in range(int(rng.integers(5, 60))):
        kind = rng.random()
        if kind < 0.75:
            n = int(rng.choice([1, 1, 1, 2, 2, 3, 4, 8, 9]))
            plan.append(("dispatch", f, n)); f += n
        elif kind < 0.82:
            plan.append(("read",))
        elif kind < 0.87:
            plan.append(("clear",)); f = 1
        elif kind < 0.92:
            plan.append(("params", int(rng.integers(0, 2))))
        else:
            plan.append(("tuning", dict(plan_every=int(rng.choice([1, 2, 8, 64])), overlap_share=int(rng.choice([-1, 0, 2, 3, 5])))))
    images = []
    for overlap in (1, 0):
        g.set_tuning(overlap_calls=overlap, plan_every=8, overlap_share=-1)
        g.set_params(params[0])
        g.clear()
        for step in plan:
            if step[0] == "dispatch":
                g.dispatch(step[1], step[2], 1); calls += 1
            elif step[0] == "read":
                g.read_image(int(rng.integers(0, 3)) if False else 0)
            elif step[0] == "clear":
                g.clear()
            elif step[0] == "params":
                g.set_params(params[step[1]])
            elif step[0] == "tuning":
                g.set_tuning(**step[1])
        images.append([g.read_image(k).tobytes() for k in range(3)])
    assert images[0] == images[1], f"run {runs}: images differ between overlapping and serial calls"
    runs += 1
    if runs % 5 == 0:
        print(f"{runs} runs, {calls} calls, all equal", flush=True)
print(f"soak done: {runs} runs, {calls} dispatch calls, overlapping == serial in every run")
