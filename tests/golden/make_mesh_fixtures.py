#!/usr/bin/env python3
"""Convert the reference's small glTF-binary meshes into raw PTM1 fixtures.

Run in the build container (needs /root/reference):
    python tests/golden/make_mesh_fixtures.py

The reference flattens every sub-mesh into a triangle soup of {pos, nor, uv}
vertices (32 B) and u32 indices, ignoring node transforms
(Source/PathTracer/PathTracingRenderer.cpp:184-446,
Source/Runtime/Asset/ModelLoader.cpp:87-88).  A PTM1 file stores exactly that
per-submesh vertex/index data, nothing else:

    char[4]  'PTM1'
    u32      vertex_count
    u32      index_count
    f32[vertex_count][8]   px py pz nx ny nz u v
    u32[index_count]

Only single-primitive meshes are converted (all of the reference's .glb models
are).  The fixtures are data; no reference source text is copied.
"""
import json
import os
import struct
import sys

import numpy as np

REF = "/root/reference/Source/PathTracer/Content/Models"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "meshes")
MODELS = ["bunny.glb", "Floor.glb", "Torus.glb", "Sphere.glb", "Teapot.glb",
          "Cube.glb", "Icosphere.glb"]

_CT = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16,
       5125: np.uint32, 5126: np.float32}
_NC = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4}


def read_glb(path):
    blob = open(path, "rb").read()
    magic, _ver, _len = struct.unpack("<III", blob[:12])
    assert magic == 0x46546C67, "not a glb"
    clen, ctype = struct.unpack("<II", blob[12:20])
    assert ctype == 0x4E4F534A
    doc = json.loads(blob[20:20 + clen])
    off = 20 + clen
    blen, btype = struct.unpack("<II", blob[off:off + 8])
    assert btype == 0x004E4942
    return doc, blob[off + 8:off + 8 + blen]


def accessor(doc, binchunk, idx):
    acc = doc["accessors"][idx]
    view = doc["bufferViews"][acc["bufferView"]]
    dt = np.dtype(_CT[acc["componentType"]])
    nc = _NC[acc["type"]]
    start = view.get("byteOffset", 0) + acc.get("byteOffset", 0)
    stride = view.get("byteStride", 0)
    count = acc["count"]
    if stride and stride != dt.itemsize * nc:
        rows = [np.frombuffer(binchunk, dt, nc, start + i * stride) for i in range(count)]
        return np.stack(rows)
    return np.frombuffer(binchunk, dt, count * nc, start).reshape(count, nc)


def convert(name):
    doc, binchunk = read_glb(os.path.join(REF, name))
    assert len(doc["meshes"]) == 1 and len(doc["meshes"][0]["primitives"]) == 1, name
    prim = doc["meshes"][0]["primitives"][0]
    pos = accessor(doc, binchunk, prim["attributes"]["POSITION"]).astype(np.float32)
    nor = accessor(doc, binchunk, prim["attributes"]["NORMAL"]).astype(np.float32)
    if "TEXCOORD_0" in prim["attributes"]:
        uv = accessor(doc, binchunk, prim["attributes"]["TEXCOORD_0"]).astype(np.float32)
    else:
        uv = np.zeros((pos.shape[0], 2), np.float32)
    idx = accessor(doc, binchunk, prim["indices"]).astype(np.uint32).reshape(-1)
    verts = np.concatenate([pos, nor, uv], axis=1).astype("<f4")
    out = os.path.join(OUT, os.path.splitext(name)[0].lower() + ".ptm")
    with open(out, "wb") as f:
        f.write(b"PTM1")
        f.write(struct.pack("<II", verts.shape[0], idx.shape[0]))
        f.write(verts.tobytes())
        f.write(idx.astype("<u4").tobytes())
    print(f"{name}: {verts.shape[0]} verts, {idx.shape[0] // 3} tris -> {out}")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    for m in MODELS:
        convert(m)
