"""The C++ example host (examples/ptrender: C ABI only, no Python in the process) against the Python
driver on the same inputs: a GLB written by the test, the same raw HDRI, the same camera.  The
float images must be identical."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

from pathtracer_amd import gltf, image_io, scenes
from pathtracer_amd import renderer as R

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PTRENDER = os.path.join(ROOT, "examples", "ptrender")


def _write_glb(path, vertices, indices, material):
    v = np.ascontiguousarray(vertices, "<f4")
    pos, nor = v[:, 0:3].copy(), v[:, 3:6].copy()
    uv = v[:, 6:8].copy()
    uv[:, 1] = np.float32(1.0) - uv[:, 1]  # the reader flips V back
    idx = np.ascontiguousarray(indices, "<u4")
    chunks, views, off = [], [], 0
    for data, target in ((pos, 34962), (nor, 34962), (uv, 34962), (idx, 34963)):
        b = data.tobytes()
        views.append({"buffer": 0, "byteOffset": off, "byteLength": len(b), "target": target})
        chunks.append(b + b"\0" * (-len(b) % 4))
        off += len(chunks[-1])
    blob = b"".join(chunks)
    n = int(v.shape[0])
    doc = {"asset": {"version": "2.0"},
           "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "NORMAL": 1, "TEXCOORD_0": 2}, "indices": 3,
                                       "material": 0}]}],
           "materials": [material],
           "accessors": [{"bufferView": 0, "componentType": 5126, "count": n, "type": "VEC3"},
                         {"bufferView": 1, "componentType": 5126, "count": n, "type": "VEC3"},
                         {"bufferView": 2, "componentType": 5126, "count": n, "type": "VEC2"},
                         {"bufferView": 3, "componentType": 5125, "count": int(idx.size), "type": "SCALAR"}],
           "bufferViews": views, "buffers": [{"byteLength": len(blob)}]}
    js = json.dumps(doc).encode()
    js += b" " * (-len(js) % 4)
    with open(path, "wb") as f:
        f.write(struct.pack("<III", 0x46546C67, 2, 12 + 8 + len(js) + 8 + len(blob)))
        f.write(struct.pack("<II", len(js), 0x4E4F534A) + js)
        f.write(struct.pack("<II", len(blob), 0x004E4942) + blob)


@pytest.mark.parametrize("device_bvh", [False, True], ids=["host-bvh", "device-bvh"])
def test_cpp_host_renders_the_same_image(tmp_path, device_bvh):
    assert os.path.exists(PTRENDER), "examples/ptrender is not built (python -c 'import __graft_entry__ as g; g.build()')"
    torus = {"pbrMetallicRoughness": {"baseColorFactor": [0.9, 0.5, 0.2, 1.0], "metallicFactor": 0.5, "roughnessFactor": 0.3},
             "extensions": {"KHR_materials_clearcoat": {"clearcoatFactor": 0.5, "clearcoatRoughnessFactor": 0.1},
                            "KHR_materials_transmission": {"transmissionFactor": 0.3}}}
    floor = {"pbrMetallicRoughness": {"baseColorFactor": [0.8, 0.8, 0.8, 1.0], "metallicFactor": 0.0},
             "extensions": {"KHR_materials_specular": {"specularFactor": 0.0}}}
    files = []
    for name, mat in (("torus", torus), ("floor", floor)):
        v, i = scenes.load_ptm(name)
        files.append(str(tmp_path / (name + ".glb")))
        _write_glb(files[-1], v, i, mat)
    hdri = scenes.sky_hdri(256, 128)
    hdri_path = str(tmp_path / "sky.f32")
    hdri.tofile(hdri_path)
    width, height, spp = 96, 64, 3
    out = str(tmp_path / "cpp.pfm")
    cmd = [PTRENDER, *files, "--width", str(width), "--height", str(height), "--spp", str(spp), "--bounces", "4",
           "--eye", "-2.0", "1.5", "0.5", "--look-at", "0", "0.2", "0", "--fovy", "1.2", "--hdri", hdri_path, "256", "128",
           "--out", out, "--ppm", str(tmp_path / "cpp.ppm")]
    if device_bvh:
        cmd.append("--device-bvh")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    cpp = image_io.read_pfm(out)

    g = R.Renderer(width, height)
    for path in files:
        gltf.add_to_renderer(g, gltf.load(path))
    g.ProcessBuffersDevice() if device_bvh else g.ProcessBuffers()
    g.upload_environment(hdri)
    p = R.default_params(width, height)
    R.set_camera(p, (-2.0, 1.5, 0.5), (0.0, 0.2, 0.0), width, height, fovy=1.2)
    p.max_bounces = 4
    g.set_params(p)
    g.clear()
    g.dispatch(1, spp, 1)
    py = g.read_image_rgb(0)
    assert cpp.shape == py.shape
    assert np.array_equal(np.ascontiguousarray(cpp).view(np.uint32), py.view(np.uint32))
    # the display image the C++ host wrote (pt_read_display, rows top to bottom in the file)
    header = f"P6\n{width} {height}\n255\n".encode()
    blob = open(tmp_path / "cpp.ppm", "rb").read()
    assert blob[:len(header)] == header and len(blob) == len(header) + width * height * 3
    ppm = np.frombuffer(blob, np.uint8, offset=len(header)).reshape(height, width, 3)[::-1]
    assert np.array_equal(ppm, g.read_display(1.0))
