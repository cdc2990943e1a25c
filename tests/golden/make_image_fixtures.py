#!/usr/bin/env python3
"""Writes tests/golden/images/c1.json (+ c1_crops.npz): the CPU oracle's output on the CPU-runnable
reference case -- BunnyClearcoat recipe on bunny + floor, 512 x 512, 3 bounces, analytic sky 1024 x 512,
FrameCount 1 and FrameCount 1..64 -- as FNV-1a-64 hashes of the raw float32 RGBA images (radiance,
albedo, normal) and 32 x 32 crops of each, so that a drift of the oracle between hosts or compilers,
and of the HIP images against the same numbers, shows without running the oracle next to it.

    python tests/golden/make_image_fixtures.py            (run in the build container; about 10 s)

The inputs are hashed too (scene buffers, HDRI, importance cache, parameter block): a mismatch there
means the inputs changed, not the renderer.  No reference output exists for this case (the reference
holds no rendered vectors and its shader cannot run here); these are the oracle's own numbers."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
W = H = 512

from tests.common import image_hash  # noqa: E402

CROPS = {"bunny": (240, 200), "floor_shadow": (330, 60), "sky_edge": (100, 400)}  # name -> (x0, y0) of a 32 x 32 window
NAMES = ("radiance", "albedo", "normal")


def c1_inputs():
    from pathtracer_amd import renderer as R
    from tests.common import built_scene, environment
    nodes, prims, mats = built_scene("BunnyClearcoat")
    hdri, cache = environment(1024, 512)
    p = R.default_params(W, H)
    return nodes, prims, mats, hdri, cache, p


def input_hashes(nodes, prims, mats, hdri, cache, p):
    import ctypes
    raw = np.frombuffer(ctypes.string_at(ctypes.addressof(p), ctypes.sizeof(p)), np.uint8)
    return {"nodes": image_hash(nodes), "primitives": image_hash(prims), "materials": image_hash(np.ascontiguousarray(mats).view(np.uint8)),
            "hdri": image_hash(hdri), "cache": image_hash(cache), "params": image_hash(raw)}


def describe(images):
    out = {}
    for name, img in zip(NAMES, images):
        out[name] = {"hash": image_hash(img), "mean_rgb": [float(img[..., k].astype(np.float64).mean()) for k in range(3)]}
    return out


def main():
    from oracle import binding as ob
    nodes, prims, mats, hdri, cache, p = c1_inputs()
    o = ob.Oracle()
    o.set_scene(nodes, prims, mats)
    o.set_environment(hdri, cache)
    threads = os.cpu_count() or 1
    one = o.render(p, W, H, 1, 1, 1, threads=threads)
    full = o.render(p, W, H, 1, 64, 1, threads=threads)
    doc = {"case": "BunnyClearcoat 512x512, 3 bounces, sky 1024x512, default camera",
           "hash": "FNV-1a-64 of every row of the array's bytes, then FNV-1a-64 of the little-endian row hashes (tests/common.py: image_hash)",
           "inputs": input_hashes(nodes, prims, mats, hdri, cache, p),
           "frame_1": describe(one[:3]), "frames_1_to_64": describe(full[:3]),
           "rays_frame_1": {"closest": int(one[4].rays_closest), "shadow": int(one[4].rays_any), "hits": int(one[4].closest_hits)},
           "rays_frames_1_to_64": {"closest": int(full[4].rays_closest), "shadow": int(full[4].rays_any), "hits": int(full[4].closest_hits)},
           "stack_overflows": int(full[4].stack_overflows), "crops": {k: list(v) for k, v in CROPS.items()}}
    crops = {}
    for tag, imgs in (("f1", one), ("f64", full)):
        for name, img in zip(NAMES, imgs[:3]):
            for cname, (x0, y0) in CROPS.items():
                crops[f"{tag}_{name}_{cname}"] = img[y0:y0 + 32, x0:x0 + 32].copy()
    out_dir = os.path.join(ROOT, "tests", "golden", "images")
    os.makedirs(out_dir, exist_ok=True)
    json.dump(doc, open(os.path.join(out_dir, "c1.json"), "w"), indent=1)
    np.savez_compressed(os.path.join(out_dir, "c1_crops.npz"), **crops)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
