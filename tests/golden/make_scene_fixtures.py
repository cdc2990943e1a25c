#!/usr/bin/env python3
"""Convert the reference's multi-material glTF scenes into .npz fixtures for the GPU parity tests.

Run in the build container (needs /root/reference and the built libptcore.so):
    python tests/golden/make_scene_fixtures.py

The scenes are read with the product's own glTF reader (pathtracer_amd.gltf), i.e. the fixtures
hold exactly what the reader hands to the renderer: per sub-mesh {pos, nor, uv} vertices and u32
indices, the 18 material attributes, the image index per texture slot, and the decoded images
(box-filtered down to at most 256 texels a side to keep the repository small; first row = bottom
of the picture).  Data only; no reference source text is copied.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pathtracer_amd import gltf  # noqa: E402

REF = "/root/reference/Source/PathTracer/Content/Models"
OUT = os.path.join(ROOT, "tests", "golden", "scenes")
SCENES = {"arealight": "AreaLight/AreaLight.gltf", "mill": "Mill/Mill.gltf"}


def convert(name, rel):
    s = gltf.load(os.path.join(REF, rel))
    arrays = {"attributes": s.attributes, "slot_images": s.slot_images}
    for k, (v, i) in enumerate(s.submeshes):
        arrays[f"v{k}"] = v
        arrays[f"i{k}"] = i
    for k in range(len(s.image_sources)):
        arrays[f"image{k}"] = s.decode_image(k, 4, max_size=256)  # RGBA; tests drop alpha where the slot is RGB
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{rel}: {len(s.submeshes)} sub-meshes, {s.triangle_count()} triangles, {len(s.image_sources)} images -> {path} "
          f"({os.path.getsize(path) / 1e6:.2f} MB)")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    for n, r in SCENES.items():
        convert(n, r)
